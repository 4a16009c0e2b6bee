"""Host-side mirror of the Open3D registration calls the reference makes outside the scan-to-map path (loop-closure
refinement, odometry constraints) over the C ABI (include/o3s_registration.h).  Names follow Open3D."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from .cloud_ops import _d


class _Criteria(C.Structure):
    _fields_ = [("relative_fitness", C.c_double), ("relative_rmse", C.c_double), ("max_iteration", C.c_int32)]


class _Result(C.Structure):
    _fields_ = [("transformation", C.c_double * 16), ("fitness", C.c_double), ("inlier_rmse", C.c_double),
                ("correspondences", C.c_int64), ("iterations", C.c_int32)]


class _Pair(C.Structure):
    _fields_ = [("source", C.POINTER(C.c_double)), ("n_source", C.c_int64), ("target", C.POINTER(C.c_double)),
                ("target_normals", C.POINTER(C.c_double)), ("n_target", C.c_int64), ("init", C.c_double * 16)]


@dataclass
class RegistrationResult:
    transformation: np.ndarray
    fitness: float
    inlier_rmse: float
    correspondences: int
    iterations: int




def _L():
    L = _lib.lib()
    if _lib.needs_binding(L, __name__):  # once per loaded library (product or test-hook build)
        dp = C.POINTER(C.c_double)
        L.o3s_o3d_registration_icp.argtypes = [C.c_int, dp, C.c_int64, dp, dp, C.c_int64, C.c_double, dp, C.POINTER(_Criteria), C.POINTER(_Result)]
        L.o3s_o3d_information_matrix.argtypes = [C.c_int, dp, C.c_int64, dp, C.c_int64, C.c_double, dp, dp]
        L.o3s_o3d_registration_icp_submaps.argtypes = [C.c_void_p, C.c_void_p, C.c_double, dp, C.POINTER(_Criteria), C.POINTER(_Result), dp]
        L.o3s_o3d_registration_icp_batch.argtypes = [C.c_int, C.c_int32, C.POINTER(_Pair), C.c_double, C.POINTER(_Criteria), C.POINTER(_Result), dp,
                                                     C.POINTER(C.c_int32)]
    return L


def _pose(T):
    return np.ascontiguousarray(np.asarray(T, np.float64).T).reshape(16)


def registration_icp(source, target, target_normals, max_correspondence_distance, init=None, relative_fitness=1e-6, relative_rmse=1e-6,
                     max_iteration=30, device: int = 0) -> RegistrationResult:
    """RegistrationICP(source, target, max_correspondence_distance, init, TransformationEstimationPointToPlane(), criteria)."""
    s_ = np.ascontiguousarray(source, np.float64)
    t_ = np.ascontiguousarray(target, np.float64)
    n_ = None if target_normals is None else np.ascontiguousarray(target_normals, np.float64)
    cr = _Criteria(float(relative_fitness), float(relative_rmse), int(max_iteration))
    r = _Result()
    rc = _L().o3s_o3d_registration_icp(device, _d(s_), s_.shape[0], _d(t_), _d(n_), t_.shape[0], float(max_correspondence_distance),
                                       _d(_pose(np.eye(4) if init is None else init)), C.byref(cr), C.byref(r))
    if rc == _lib.ERR_BAD_SHAPE:
        raise RuntimeError("TransformationEstimationPointToPlane requires target normals")
    if rc != _lib.OK:
        raise RuntimeError(f"o3s_o3d_registration_icp failed with o3s_status {rc}")
    return RegistrationResult(np.array(r.transformation).reshape(4, 4).T.copy(), r.fitness, r.inlier_rmse, int(r.correspondences), int(r.iterations))


def get_information_matrix_from_point_clouds(source, target, max_correspondence_distance, transformation, device: int = 0) -> np.ndarray:
    s_ = np.ascontiguousarray(source, np.float64)
    t_ = np.ascontiguousarray(target, np.float64)
    out = np.zeros(36)
    rc = _L().o3s_o3d_information_matrix(device, _d(s_), s_.shape[0], _d(t_), t_.shape[0], float(max_correspondence_distance),
                                         _d(_pose(transformation)), _d(out))
    if rc != _lib.OK:
        raise RuntimeError(f"o3s_o3d_information_matrix failed with o3s_status {rc}")
    return out.reshape(6, 6).T.copy()


def registration_icp_batch(pairs, max_correspondence_distance, relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=30,
                           with_information: bool = False, device: int = 0):
    """RegistrationICP over independent candidate pairs, run concurrently on one device (the loop-closure candidates of
    PlaceRecognition.cpp:70-150).  pairs: sequence of (source, target, target_normals, init | None).
    Returns [RegistrationResult] — and [6x6 information matrix] with with_information."""
    n = len(pairs)
    if n == 0:
        return ([], []) if with_information else []
    keep = []  # the arrays the structs point into
    arr = (_Pair * n)()
    for k, (src, tgt, tn, init) in enumerate(pairs):
        s_ = np.ascontiguousarray(src, np.float64)
        t_ = np.ascontiguousarray(tgt, np.float64)
        if tn is None:
            raise RuntimeError("TransformationEstimationPointToPlane requires target normals")
        n_ = np.ascontiguousarray(tn, np.float64)
        keep += [s_, t_, n_]
        arr[k].source, arr[k].n_source = _d(s_), s_.shape[0]
        arr[k].target, arr[k].target_normals, arr[k].n_target = _d(t_), _d(n_), t_.shape[0]
        arr[k].init = (C.c_double * 16)(*_pose(np.eye(4) if init is None else init))
    cr = _Criteria(float(relative_fitness), float(relative_rmse), int(max_iteration))
    res = (_Result * n)()
    status = (C.c_int32 * n)()
    infos = np.zeros((n, 36)) if with_information else None
    rc = _L().o3s_o3d_registration_icp_batch(device, n, arr, float(max_correspondence_distance), C.byref(cr), res, _d(infos), status)
    if rc != _lib.OK:
        raise RuntimeError(f"o3s_o3d_registration_icp_batch failed with o3s_status {rc} (per pair: {list(status)})")
    out = [RegistrationResult(np.array(r.transformation).reshape(4, 4).T.copy(), r.fitness, r.inlier_rmse, int(r.correspondences), int(r.iterations))
           for r in res]
    if with_information:
        return out, [infos[k].reshape(6, 6).T.copy() for k in range(n)]
    return out


def registration_icp_submaps(source_submap, target_submap, max_correspondence_distance, init=None, relative_fitness=1e-6, relative_rmse=1e-6,
                             max_iteration=30, with_information: bool = False):
    """RegistrationICP between the map clouds of two device-resident Submap objects (constraint_builders.cpp:55-75,
    PlaceRecognition.cpp:111): neither cloud leaves HBM.  Returns RegistrationResult (and the 6x6 information matrix)."""
    cr = _Criteria(float(relative_fitness), float(relative_rmse), int(max_iteration))
    r = _Result()
    info = np.zeros(36) if with_information else None
    rc = _L().o3s_o3d_registration_icp_submaps(source_submap._h, target_submap._h, float(max_correspondence_distance),
                                               _d(_pose(np.eye(4) if init is None else init)), C.byref(cr), C.byref(r), _d(info))
    if rc == _lib.ERR_BAD_SHAPE:
        raise RuntimeError("TransformationEstimationPointToPlane requires target normals")
    if rc != _lib.OK:
        raise RuntimeError(f"o3s_o3d_registration_icp_submaps failed with o3s_status {rc}")
    res = RegistrationResult(np.array(r.transformation).reshape(4, 4).T.copy(), r.fitness, r.inlier_rmse, int(r.correspondences), int(r.iterations))
    return (res, info.reshape(6, 6).T.copy()) if with_information else res


def compute_indices_of_overlapping_points(source, target, source_to_target, voxel_size, min_num_points_per_voxel: int = 1, device: int = 0):
    """computeIndicesOfOverlappingPoints (open3d_slam/src/helpers.cpp:319-345): (idxsSource, idxsTarget), ascending."""
    s_ = np.ascontiguousarray(source, np.float64)
    t_ = np.ascontiguousarray(target, np.float64)
    i_s = np.zeros(max(s_.shape[0], 1), np.int64)
    i_t = np.zeros(max(t_.shape[0], 1), np.int64)
    ns, nt = C.c_int64(), C.c_int64()
    L = _L()
    ip = C.POINTER(C.c_int64)
    L.o3s_overlap_indices.argtypes = [C.c_int, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double), C.c_double,
                                      C.c_int64, ip, ip, ip, ip]
    rc = L.o3s_overlap_indices(device, _d(s_), s_.shape[0], _d(t_), t_.shape[0], _d(_pose(source_to_target)), float(voxel_size),
                               int(min_num_points_per_voxel), i_s.ctypes.data_as(ip), C.byref(ns), i_t.ctypes.data_as(ip), C.byref(nt))
    if rc != _lib.OK:
        raise RuntimeError(f"o3s_overlap_indices failed with o3s_status {rc}")
    return i_s[:ns.value].copy(), i_t[:nt.value].copy()


def registration_icp_submaps_overlap(source_submap, target_submap, max_correspondence_distance, init, overlap_voxel_size,
                                     min_num_points_per_voxel: int = 1, relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=30,
                                     with_information: bool = True):
    """The loop-closure refinement of PlaceRecognition::buildLoopClosureConstraints (PlaceRecognition.cpp:97-150) between two
    device-resident Submap objects: overlap selection at `init`, RegistrationICP on the selections, information matrix.
    Returns (RegistrationResult, information 6x6 or None, (n_source_overlap, n_target_overlap)); an empty overlap gives None."""
    cr = _Criteria(float(relative_fitness), float(relative_rmse), int(max_iteration))
    r = _Result()
    info = np.zeros(36) if with_information else None
    n_ov = (C.c_int64 * 2)()
    L = _L()
    L.o3s_o3d_registration_icp_submaps_overlap.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.POINTER(C.c_double), C.POINTER(_Criteria), C.c_double,
                                                           C.c_int64, C.POINTER(_Result), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    rc = L.o3s_o3d_registration_icp_submaps_overlap(source_submap._h, target_submap._h, float(max_correspondence_distance), _d(_pose(init)),
                                                    C.byref(cr), float(overlap_voxel_size), int(min_num_points_per_voxel), C.byref(r), _d(info), n_ov)
    if rc == _lib.ERR_EMPTY_REFERENCE:
        return None, None, (int(n_ov[0]), int(n_ov[1]))
    if rc == _lib.ERR_BAD_SHAPE:
        raise RuntimeError("TransformationEstimationPointToPlane requires target normals")
    if rc != _lib.OK:
        raise RuntimeError(f"o3s_o3d_registration_icp_submaps_overlap failed with o3s_status {rc}")
    res = RegistrationResult(np.array(r.transformation).reshape(4, 4).T.copy(), r.fitness, r.inlier_rmse, int(r.correspondences), int(r.iterations))
    return res, (info.reshape(6, 6).T.copy() if with_information else None), (int(n_ov[0]), int(n_ov[1]))


def registration_icp_submaps_overlap_batch(pairs, max_correspondence_distance, overlap_voxel_size, min_num_points_per_voxel: int = 1,
                                           relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=30):
    """o3s_o3d_registration_icp_submaps_overlap_batch: the loop-closure refinement for several (source Submap, target Submap, init)
    triples at once, up to four in flight on the device.  Returns a list of (RegistrationResult | None, information 6x6 | None,
    (n_source_overlap, n_target_overlap), status) — None for a pair whose overlap is empty."""
    n = len(pairs)
    if n == 0:
        return []
    cr = _Criteria(float(relative_fitness), float(relative_rmse), int(max_iteration))
    srcs = (C.c_void_p * n)(*[p_[0]._h for p_ in pairs])
    tgts = (C.c_void_p * n)(*[p_[1]._h for p_ in pairs])
    inits = np.ascontiguousarray(np.stack([_pose(p_[2]) for p_ in pairs]), np.float64)
    res = (_Result * n)()
    infos = np.zeros((n, 36))
    novs = (C.c_int64 * (2 * n))()
    sts = (C.c_int32 * n)()
    L = _L()
    L.o3s_o3d_registration_icp_submaps_overlap_batch.argtypes = [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_double, C.POINTER(C.c_double),
                                                                 C.POINTER(_Criteria), C.c_double, C.c_int64, C.POINTER(_Result), C.POINTER(C.c_double),
                                                                 C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    rc = L.o3s_o3d_registration_icp_submaps_overlap_batch(n, srcs, tgts, float(max_correspondence_distance), _d(inits), C.byref(cr), float(overlap_voxel_size),
                                                          int(min_num_points_per_voxel), res, _d(infos), novs, sts)
    if rc != _lib.OK:
        raise RuntimeError(f"o3s_o3d_registration_icp_submaps_overlap_batch failed with o3s_status {rc}")
    out = []
    for k in range(n):
        nov = (int(novs[2 * k]), int(novs[2 * k + 1]))
        if sts[k] != _lib.OK:
            out.append((None, None, nov, int(sts[k])))
            continue
        r = res[k]
        out.append((RegistrationResult(np.array(r.transformation).reshape(4, 4).T.copy(), r.fitness, r.inlier_rmse, int(r.correspondences), int(r.iterations)),
                    infos[k].reshape(6, 6).T.copy(), nov, 0))
    return out


def reserve(max_source_points: int, max_target_points: int, device: int = 0):
    """o3s_o3d_registration_reserve: sizes the device's registration work area once, so that no registration of clouds up to these
    sizes allocates (an allocation stalls every stream of the device for milliseconds)."""
    L = _L()
    L.o3s_o3d_registration_reserve.argtypes = [C.c_int, C.c_int64, C.c_int64]
    rc = L.o3s_o3d_registration_reserve(int(device), int(max_source_points), int(max_target_points))
    if rc != _lib.OK:
        raise RuntimeError(f"o3s_o3d_registration_reserve failed with o3s_status {rc}")


def reserve_n(max_source_points: int, max_target_points: int, count: int, device: int = 0):
    """o3s_o3d_registration_reserve_n: `count` work areas of that size (the lanes of the batch entries)."""
    L = _L()
    L.o3s_o3d_registration_reserve_n.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int32]
    rc = L.o3s_o3d_registration_reserve_n(int(device), int(max_source_points), int(max_target_points), int(count))
    if rc != _lib.OK:
        raise RuntimeError(f"o3s_o3d_registration_reserve_n failed with o3s_status {rc}")


def release(device: int = 0):
    """o3s_o3d_registration_release: the idle registration work areas of the device go back to the allocator."""
    L = _L()
    L.o3s_o3d_registration_release.argtypes = [C.c_int]
    rc = L.o3s_o3d_registration_release(int(device))
    if rc != _lib.OK:
        raise RuntimeError(f"o3s_o3d_registration_release failed with o3s_status {rc}")
