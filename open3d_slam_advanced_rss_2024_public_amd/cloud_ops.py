"""Host-side mirror of the open3d_slam point-cloud helpers around the ICP path, over the C ABI
(include/o3s_cloud_ops.h).  Function names follow the reference (open3d_slam/src/helpers.cpp, croppers.cpp,
include/open3d_slam/VoxelHashMap.hpp, open3d_conversions.cpp); all arithmetic runs on the GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


class CropperC(C.Structure):
    _fields_ = [("kind", C.c_int32), ("invert", C.c_int32), ("p0", C.c_double), ("p1", C.c_double), ("p2", C.c_double),
                ("centre", C.c_double * 3)]


_KINDS = {"CroppingVolume": 0, "MaxRadius": 1, "MinRadius": 2, "MinMaxRadius": 3, "Cylinder": 4}


def _L():
    L = _lib.lib()
    if _lib.needs_binding(L, __name__):  # once per loaded library (product or test-hook build)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        fp = C.POINTER(C.c_float)
        L.o3s_voxel_idx.argtypes = [C.c_int, dp, C.c_int64, C.c_double, ip]
        L.o3s_voxel_hash.argtypes = [C.c_int, ip, C.c_int64, C.POINTER(C.c_uint64)]
        L.o3s_crop.argtypes = [C.c_int, C.POINTER(CropperC), dp, dp, C.c_int64, dp, dp, C.POINTER(C.c_int64)]
        L.o3s_voxelize_within_crop.argtypes = [C.c_int, C.POINTER(CropperC), C.c_double, dp, dp, C.c_int64, dp, dp, ip,
                                               C.POINTER(C.c_int64)]
        L.o3s_voxel_downsample.argtypes = [C.c_int, C.c_double, dp, dp, C.c_int64, dp, dp, ip, C.POINTER(C.c_int64)]
        L.o3s_o3d_to_pm.argtypes = [C.c_int, dp, dp, C.c_int64, fp, fp]
        L.o3s_estimate_normals.argtypes = [C.c_int, dp, C.c_int64, C.c_double, C.c_int32, dp, ip]
    return L


def _check(rc, what):
    if rc != _lib.OK:
        raise RuntimeError(f"{what} failed with o3s_status {rc}")


def _d(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


def croppingVolumeFactory(kind: str = "MaxRadius", p0=0.0, p1=0.0, p2=0.0, centre=(0.0, 0.0, 0.0), invert=False) -> CropperC:
    """croppingVolumeFactory (croppers.cpp:14-51) + setPose / setIsInvertVolume."""
    return CropperC(_KINDS[kind], int(invert), float(p0), float(p1), float(p2), (C.c_double * 3)(*[float(v) for v in centre]))


def getVoxelIdx(points, voxel_size: float, device: int = 0) -> np.ndarray:
    p = np.ascontiguousarray(points, np.float64)
    out = np.zeros((p.shape[0], 3), np.int32)
    _check(_L().o3s_voxel_idx(device, _d(p), p.shape[0], float(voxel_size), _i(out)), "o3s_voxel_idx")
    return out


def voxelHash(idx, device: int = 0) -> np.ndarray:
    i = np.ascontiguousarray(idx, np.int32)
    out = np.zeros(i.shape[0], np.uint64)
    _check(_L().o3s_voxel_hash(device, _i(i), i.shape[0], out.ctypes.data_as(C.POINTER(C.c_uint64))), "o3s_voxel_hash")
    return out


def crop(cropper: CropperC, points, normals=None, device: int = 0):
    """CroppingVolume::crop (croppers.cpp:76-106)."""
    p = np.ascontiguousarray(points, np.float64)
    n = None if normals is None else np.ascontiguousarray(normals, np.float64)
    op = np.zeros_like(p)
    on = np.zeros_like(p) if n is not None else None
    k = C.c_int64()
    _check(_L().o3s_crop(device, C.byref(cropper), _d(p), _d(n), p.shape[0], _d(op), _d(on), C.byref(k)), "o3s_crop")
    return op[:k.value].copy(), (on[:k.value].copy() if n is not None else None)


def voxelizeWithinCroppingVolume(voxel_size: float, cropper: CropperC, points, normals=None, device: int = 0):
    """voxelizeWithinCroppingVolume (helpers.cpp:117-192) -> (points, normals, voxel_idx)."""
    p = np.ascontiguousarray(points, np.float64)
    n = None if normals is None else np.ascontiguousarray(normals, np.float64)
    op = np.zeros_like(p)
    on = np.zeros_like(p) if n is not None else None
    oi = np.zeros((p.shape[0], 3), np.int32)
    k = C.c_int64()
    _check(_L().o3s_voxelize_within_crop(device, C.byref(cropper), float(voxel_size), _d(p), _d(n), p.shape[0], _d(op), _d(on),
                                         _i(oi), C.byref(k)), "o3s_voxelize_within_crop")
    return op[:k.value].copy(), (on[:k.value].copy() if n is not None else None), oi[:k.value].copy()


def voxelize(voxel_size: float, points, normals=None, device: int = 0):
    """o3d_slam::voxelize (helpers.cpp:108-115) = Open3D v0.15.1 VoxelDownSample -> (points, normals, voxel_idx)."""
    p = np.ascontiguousarray(points, np.float64)
    n = None if normals is None else np.ascontiguousarray(normals, np.float64)
    op = np.zeros_like(p)
    on = np.zeros_like(p) if n is not None else None
    oi = np.zeros((p.shape[0], 3), np.int32)
    k = C.c_int64()
    _check(_L().o3s_voxel_downsample(device, float(voxel_size), _d(p), _d(n), p.shape[0], _d(op), _d(on), _i(oi), C.byref(k)),
           "o3s_voxel_downsample")
    return op[:k.value].copy(), (on[:k.value].copy() if n is not None else None), oi[:k.value].copy()


def open3dToPointmatcher(points, normals=None, device: int = 0):
    """open3dToPointmatcher (open3d_conversions.cpp:57-118) -> (xyzw (N,4) fp32, normals (N,3) fp32 | None)."""
    p = np.ascontiguousarray(points, np.float64)
    n = None if normals is None else np.ascontiguousarray(normals, np.float64)
    xyzw = np.zeros((p.shape[0], 4), np.float32)
    on = np.zeros((p.shape[0], 3), np.float32) if n is not None else None
    fp = C.POINTER(C.c_float)
    _check(_L().o3s_o3d_to_pm(device, _d(p), _d(n), p.shape[0], xyzw.ctypes.data_as(fp), None if on is None else on.ctypes.data_as(fp)),
           "o3s_o3d_to_pm")
    return xyzw, on


def estimateNormals(points, max_radius: float, knn: int, want_neighbours: bool = False, device: int = 0):
    """EstimateNormals(KDTreeSearchParamHybrid(max_radius, knn)) + NormalizeNormals + OrientNormalsTowardsCameraLocation
    (CloudRegistration.cpp:71-74)."""
    p = np.ascontiguousarray(points, np.float64)
    out = np.zeros_like(p)
    nn = np.zeros((p.shape[0], knn), np.int32) if want_neighbours else None
    _check(_L().o3s_estimate_normals(device, _d(p), p.shape[0], float(max_radius), int(knn), _d(out), _i(nn)), "o3s_estimate_normals")
    return (out, nn) if want_neighbours else out


# ---- the same operators with colours / covariances riding along (include/o3s_cloud_ops.h, *_attr entry points) ----------
def _attr_call(fn_name, lead_args, points, normals, colors, covariances, want_idx, device):
    p = np.ascontiguousarray(points, np.float64)
    n = None if normals is None else np.ascontiguousarray(normals, np.float64)
    col = None if colors is None else np.ascontiguousarray(colors, np.float64)
    cov = None if covariances is None else np.ascontiguousarray(covariances, np.float64).reshape(-1, 9)
    op = np.zeros_like(p)
    on = np.zeros_like(p) if n is not None else None
    oc = np.zeros_like(p) if col is not None else None
    ov = np.zeros((p.shape[0], 9)) if cov is not None else None
    oi = np.zeros((p.shape[0], 3), np.int32) if want_idx else None
    k = C.c_int64()
    L = _L()
    fn = getattr(L, fn_name)
    dp = C.POINTER(C.c_double)
    args = [device] + lead_args + [_d(p), _d(n), _d(col), _d(cov), C.c_int64(p.shape[0]), _d(op), _d(on), _d(oc), _d(ov)]
    if want_idx:
        args.append(_i(oi))
    args.append(C.byref(k))
    fn.restype = C.c_int
    fn.argtypes = None
    _check(fn(*args), fn_name)
    m = k.value
    out = [op[:m].copy(), None if on is None else on[:m].copy(), None if oc is None else oc[:m].copy(), None if ov is None else ov[:m].copy()]
    if want_idx:
        out.append(oi[:m].copy())
    return tuple(out)


def crop_attr(cropper: CropperC, points, normals=None, colors=None, covariances=None, device: int = 0):
    """CroppingVolume::crop with colours and covariances (croppers.cpp:76-106) -> (points, normals, colors, covariances)."""
    return _attr_call("o3s_crop_attr", [C.byref(cropper)], points, normals, colors, covariances, False, device)


def voxelizeWithinCroppingVolume_attr(voxel_size: float, cropper: CropperC, points, normals=None, colors=None, covariances=None, device: int = 0):
    """voxelizeWithinCroppingVolume with colours (last colour of the voxel) and covariances (mean) -> (p, n, col, cov, voxel_idx)."""
    return _attr_call("o3s_voxelize_within_crop_attr", [C.byref(cropper), C.c_double(float(voxel_size))], points, normals, colors, covariances, True, device)


def voxelize_attr(voxel_size: float, points, normals=None, colors=None, covariances=None, device: int = 0):
    """Open3D VoxelDownSample with mean colours / covariances -> (p, n, col, cov, voxel_idx)."""
    return _attr_call("o3s_voxel_downsample_attr", [C.c_double(float(voxel_size))], points, normals, colors, covariances, True, device)


def transform(T, points, normals=None, covariances=None, device: int = 0):
    """o3d_slam::transform (helpers.cpp:283-318) -> (points, normals, covariances); an (almost-)identity T doubles the cloud."""
    p = np.ascontiguousarray(points, np.float64)
    n = None if normals is None else np.ascontiguousarray(normals, np.float64)
    cov = None if covariances is None else np.ascontiguousarray(covariances, np.float64).reshape(-1, 9)
    Tc = np.ascontiguousarray(np.asarray(T, np.float64).T).reshape(16)
    op = np.zeros((2 * p.shape[0], 3))
    on = np.zeros((2 * p.shape[0], 3)) if n is not None else None
    ov = np.zeros((2 * p.shape[0], 9)) if cov is not None else None
    k = C.c_int64()
    L = _L()
    L.o3s_transform_cloud.restype = C.c_int
    L.o3s_transform_cloud.argtypes = None
    _check(L.o3s_transform_cloud(device, _d(Tc), _d(p), _d(n), _d(cov), C.c_int64(p.shape[0]), _d(op), _d(on), _d(ov), C.byref(k)), "o3s_transform_cloud")
    m = k.value
    return op[:m].copy(), (None if on is None else on[:m].copy()), (None if ov is None else ov[:m].copy())
