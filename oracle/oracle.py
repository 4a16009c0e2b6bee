"""ctypes front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

OK = 0
ERR_EMPTY_REFERENCE = 1
ERR_EMPTY_READING = 2
ERR_BAD_SHAPE = 3
ERR_NOT_INITIALIZED = 4
ERR_NO_MATCHES = 5
ERR_NO_POINTS = 6
ERR_NAN = 7
ERR_NOT_RIGID = 8
ERR_BAD_CONFIG = 9


class _Cfg(C.Structure):
    _fields_ = [
        ("matcher", C.c_int32),
        ("max_dist", C.c_float),
        ("trim_ratio", C.c_float),
        ("max_normal_angle", C.c_float),
        ("max_dist_outlier", C.c_float),
        ("use_differential", C.c_int32),
        ("min_diff_rot", C.c_float),
        ("min_diff_trans", C.c_float),
        ("smooth_length", C.c_int32),
        ("max_iters", C.c_int32),
        ("counter_first", C.c_int32),
    ]


class _Stats(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32),
        ("max_iters_reached", C.c_int32),
        ("kept_pairs", C.c_int64),
        ("matched_pairs", C.c_int64),
        ("point_used_ratio", C.c_float),
        ("weighted_point_used_ratio", C.c_float),
        ("last_trim_limit", C.c_float),
        ("match_ms", C.c_double),
        ("outlier_ms", C.c_double),
        ("minimize_ms", C.c_double),
        ("total_ms", C.c_double),
    ]


class _O3dIcpResult(C.Structure):
    _fields_ = [("transformation", C.c_double * 16), ("fitness", C.c_double), ("inlier_rmse", C.c_double),
                ("correspondences", C.c_int64), ("iterations", C.c_int32)]


class _Cropper(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("invert", C.c_int32),
        ("p0", C.c_double),
        ("p1", C.c_double),
        ("p2", C.c_double),
        ("centre", C.c_double * 3),
    ]


def build(force: bool = False) -> str:
    """Compile oracle/liboracle.so with g++ (oracle/Makefile)."""
    srcs = ["icp_oracle.cpp", "icp_oracle.h", "dense_map_oracle.cpp", "dense_map_oracle.h"]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
        os.path.getmtime(os.path.join(_HERE, f)) for f in srcs
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(_Cfg)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_nabo_epsilon.argtypes = [C.c_void_p, C.c_float]
        L.orc_set_nabo_epsilon.restype = None
        L.orc_init_reference.argtypes = [C.c_void_p, fp, fp, C.c_int64]
        L.orc_matcher_init.argtypes = [C.c_void_p, fp, fp, C.c_int64]
        L.orc_compute.argtypes = [C.c_void_p, fp, fp, C.c_int64, fp, fp, C.POINTER(_Stats), fp, fp,
                                  C.POINTER(C.c_int64), C.c_int32]
        L.orc_reference_mean.argtypes = [C.c_void_p, fp]
        L.orc_find_closests.argtypes = [C.c_void_p, fp, C.c_int64, ip, fp, C.c_int]
        L.orc_dists_quantile.argtypes = [fp, C.c_int64, C.c_float, fp]
        L.orc_outlier_weights.argtypes = [C.c_void_p, fp, ip, fp, C.c_int64, fp]
        L.orc_p2plane_step.argtypes = [C.c_void_p, fp, ip, fp, fp, C.c_int64, fp, fp, fp, fp]
        L.orc_solve6.argtypes = [fp, fp, fp, ip]
        L.orc_rigid_transform.argtypes = [fp, fp, fp, C.c_int64]
        L.orc_voxel_idx.argtypes = [dp, C.c_int64, C.c_double, ip]
        L.orc_voxel_idx_div.argtypes = [dp, C.c_int64, C.c_double, dp, ip]
        L.orc_voxel_hash.argtypes = [ip, C.c_int64, C.POINTER(C.c_uint64)]
        L.orc_crop_mask.argtypes = [C.POINTER(_Cropper), dp, C.c_int64, C.POINTER(C.c_uint8)]
        L.orc_voxelize_within_crop.restype = C.c_int64
        L.orc_voxelize_within_crop.argtypes = [C.POINTER(_Cropper), C.c_double, dp, dp, C.c_int64, dp, dp, ip]
        L.orc_voxel_downsample_o3d.restype = C.c_int64
        L.orc_voxel_downsample_o3d.argtypes = [C.c_double, dp, dp, C.c_int64, dp, dp, ip]
        L.orc_o3d_to_pm.argtypes = [dp, dp, C.c_int64, fp, fp]
        L.orc_estimate_normals.argtypes = [dp, C.c_int64, C.c_double, C.c_int32, dp, ip]
        L.orc_o3d_registration_icp.argtypes = [dp, C.c_int64, dp, dp, C.c_int64, C.c_double, dp, C.c_double, C.c_double, C.c_int32,
                                               C.POINTER(_O3dIcpResult)]
        L.orc_o3d_information_matrix.argtypes = [dp, C.c_int64, dp, C.c_int64, C.c_double, dp, dp]
        L.orc_carve.restype = None
        L.orc_carve.argtypes = [dp, C.c_int64, dp, dp, C.c_int64, C.POINTER(C.c_uint8), dp, C.c_double, C.c_double, C.c_double, C.c_double,
                                C.POINTER(C.c_uint8)]
        L.orc_transform_cloud.restype = C.c_int64
        L.orc_transform_cloud.argtypes = [dp, dp, dp, C.c_int64, dp, dp]
        u8p = C.POINTER(C.c_uint8)
        L.orc_dense_create.restype = C.c_void_p
        L.orc_dense_create.argtypes = [C.c_double]
        L.orc_dense_destroy.argtypes = [C.c_void_p]
        L.orc_dense_size.restype = C.c_int64
        L.orc_dense_size.argtypes = [C.c_void_p]
        L.orc_dense_insert.argtypes = [C.c_void_p, dp, dp, C.c_int64]
        L.orc_dense_to_point_cloud.restype = C.c_int64
        L.orc_dense_to_point_cloud.argtypes = [C.c_void_p, dp, dp, ip, ip]
        L.orc_dense_transform.argtypes = [C.c_void_p, dp]
        L.orc_remove_duplicate_points.restype = C.c_int64
        L.orc_remove_duplicate_points.argtypes = [dp, C.c_int64, C.c_double, u8p]
        L.orc_voxels_within_neighborhood.restype = C.c_int64
        L.orc_voxels_within_neighborhood.argtypes = [dp, C.c_double, C.c_double, ip, C.c_int64]
        L.orc_dense_carve.restype = C.c_int64
        L.orc_dense_carve.argtypes = [C.c_void_p, dp, C.c_int64, dp, C.c_double, C.c_double, C.c_double]
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def as_xyzw(points: np.ndarray) -> np.ndarray:
    """(N,3) or (N,4) array -> contiguous (N,4) float32 == PM features.data() of a 4xN column-major matrix."""
    p = np.asarray(points, dtype=np.float32)
    if p.shape[1] == 3:
        p = np.concatenate([p, np.ones((p.shape[0], 1), np.float32)], axis=1)
    return np.ascontiguousarray(p, dtype=np.float32)


def as_normals(n):
    return None if n is None else np.ascontiguousarray(n, dtype=np.float32)


def mat_to_colmajor(T: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).T).reshape(16)


def colmajor_to_mat(t: np.ndarray) -> np.ndarray:
    return np.asarray(t, dtype=np.float32).reshape(4, 4).T.copy()


@dataclass
class OracleConfig:
    """Mirror of an ICP yaml chain (open3d_slam_ros/param/icp.yaml defaults)."""
    matcher: int = 0               # 0 KDTreeMatcher, 1 MirrorMatcher
    max_dist: float = 0.5
    trim_ratio: float = 0.9        # < 0: filter absent
    max_normal_angle: float = 1.57  # < 0: filter absent
    max_dist_outlier: float = -1.0
    use_differential: bool = True
    min_diff_rot: float = 0.001
    min_diff_trans: float = 0.01
    smooth_length: int = 3
    max_iters: int = 15
    counter_first: bool = False

    def to_c(self) -> _Cfg:
        return _Cfg(self.matcher, self.max_dist, self.trim_ratio, self.max_normal_angle, self.max_dist_outlier,
                    int(self.use_differential), self.min_diff_rot, self.min_diff_trans, self.smooth_length,
                    self.max_iters, int(self.counter_first))


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__(f"oracle status {code}")
        self.code = code


class OracleIcp:
    def __init__(self, cfg: OracleConfig, threads: int = 1):
        self.cfg = cfg
        c = cfg.to_c()
        self._h = lib().orc_create(C.byref(c))
        lib().orc_set_threads(self._h, threads)
        self.trace_T = None
        self.trace_limit = None
        self.trace_kept = None
        self.stats = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_destroy(self._h)
            self._h = None

    def set_threads(self, n):
        lib().orc_set_threads(self._h, n)

    def set_nabo_epsilon(self, epsilon: float):
        """epsilon >= 0: libnabo's epsilon-approximate KDTREE_LINEAR_HEAP search (restated); < 0: the exact search."""
        lib().orc_set_nabo_epsilon(self._h, float(epsilon))

    def init_reference(self, xyz, normals) -> int:
        xyzw = as_xyzw(xyz)
        nn = as_normals(normals)
        return lib().orc_init_reference(self._h, _f(xyzw), _f(nn), xyzw.shape[0])

    def matcher_init(self, xyz, normals=None) -> int:
        """Matcher::init (LPM/MatchersImpl.cpp:108-114): the cloud indexed as given, no mean subtraction."""
        xyzw = as_xyzw(xyz)
        nn = as_normals(normals)
        return lib().orc_matcher_init(self._h, _f(xyzw), _f(nn), xyzw.shape[0])

    def reference_mean(self):
        m = np.zeros(3, np.float32)
        lib().orc_reference_mean(self._h, _f(m))
        return m

    def compute(self, xyz, normals, T_init, raise_on_error=True):
        xyzw = as_xyzw(xyz)
        nn = as_normals(normals)
        cap = max(self.cfg.max_iters, 1) if self.cfg.max_iters > 0 else 1024
        tT = np.zeros((cap, 16), np.float32)
        tl = np.zeros(cap, np.float32)
        tk = np.zeros(cap, np.int64)
        st = _Stats()
        Tin = mat_to_colmajor(T_init)
        Tout = np.zeros(16, np.float32)
        code = lib().orc_compute(self._h, _f(xyzw), _f(nn), xyzw.shape[0], _f(Tin), _f(Tout), C.byref(st), _f(tT),
                                 _f(tl), tk.ctypes.data_as(C.POINTER(C.c_int64)), cap)
        self.stats = st
        it = st.iterations
        self.trace_T = np.stack([colmajor_to_mat(t) for t in tT[:it]]) if it else np.zeros((0, 4, 4), np.float32)
        self.trace_limit = tl[:it].copy()
        self.trace_kept = tk[:it].copy()
        if code != OK:
            if raise_on_error:
                raise OracleError(code)
            return None, code
        return (colmajor_to_mat(Tout), code) if not raise_on_error else colmajor_to_mat(Tout)

    # ---- module level ----
    def find_closests(self, query_xyz, brute=False):
        q = as_xyzw(query_xyz)
        n = q.shape[0]
        ids = np.zeros(n, np.int32)
        d2 = np.zeros(n, np.float32)
        code = lib().orc_find_closests(self._h, _f(q), n, _i(ids), _f(d2), int(brute))
        if code != OK:
            raise OracleError(code)
        return ids, d2

    def outlier_weights(self, reading_normals, ids, d2):
        nn = as_normals(reading_normals)
        ids = np.ascontiguousarray(ids, np.int32)
        d2 = np.ascontiguousarray(d2, np.float32)
        w = np.zeros(ids.shape[0], np.float32)
        code = lib().orc_outlier_weights(self._h, _f(nn), _i(ids), _f(d2), ids.shape[0], _f(w))
        if code != OK:
            raise OracleError(code)
        return w

    def p2plane_step(self, reading_xyz, ids, d2, w):
        q = as_xyzw(reading_xyz)
        ids = np.ascontiguousarray(ids, np.int32)
        d2 = np.ascontiguousarray(d2, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        T = np.zeros(16, np.float32)
        A = np.zeros(36, np.float32)
        b = np.zeros(6, np.float32)
        x = np.zeros(6, np.float32)
        code = lib().orc_p2plane_step(self._h, _f(q), _i(ids), _f(d2), _f(w), q.shape[0], _f(T), _f(A), _f(b), _f(x))
        if code != OK:
            raise OracleError(code)
        return colmajor_to_mat(T), A.reshape(6, 6).T.copy(), b, x


def dists_quantile(d2, ratio):
    d2 = np.ascontiguousarray(d2, np.float32)
    out = C.c_float()
    code = lib().orc_dists_quantile(_f(d2), d2.shape[0], ratio, C.byref(out))
    if code != OK:
        raise OracleError(code)
    return np.float32(out.value)


def solve6(A, b):
    Ac = np.ascontiguousarray(np.asarray(A, np.float32).T).reshape(36)
    bc = np.ascontiguousarray(b, np.float32)
    x = np.zeros(6, np.float32)
    br = C.c_int32()
    lib().orc_solve6(_f(Ac), _f(bc), _f(x), C.byref(br))
    return x, br.value


def rigid_transform(T, xyz, normals=None):
    q = as_xyzw(xyz).copy()
    nn = None if normals is None else as_normals(normals).copy()
    Tc = mat_to_colmajor(T)
    code = lib().orc_rigid_transform(_f(Tc), _f(q), _f(nn), q.shape[0])
    if code != OK:
        raise OracleError(code)
    return q[:, :3].copy(), nn


def voxel_idx(pts, voxel_size):
    p = np.ascontiguousarray(pts, np.float64)
    idx = np.zeros((p.shape[0], 3), np.int32)
    lib().orc_voxel_idx(_d(p), p.shape[0], float(voxel_size), _i(idx))
    return idx


def voxel_idx_div(pts, voxel_size, min_bound=None):
    p = np.ascontiguousarray(pts, np.float64)
    mb = None if min_bound is None else np.ascontiguousarray(min_bound, np.float64)
    idx = np.zeros((p.shape[0], 3), np.int32)
    lib().orc_voxel_idx_div(_d(p), p.shape[0], float(voxel_size), _d(mb), _i(idx))
    return idx


def voxel_hash(idx):
    i = np.ascontiguousarray(idx, np.int32)
    h = np.zeros(i.shape[0], np.uint64)
    lib().orc_voxel_hash(_i(i), i.shape[0], h.ctypes.data_as(C.POINTER(C.c_uint64)))
    return h


def make_cropper(kind="MaxRadius", p0=0.0, p1=0.0, p2=0.0, centre=(0, 0, 0), invert=False) -> _Cropper:
    kinds = {"Base": 0, "MaxRadius": 1, "MinRadius": 2, "MinMaxRadius": 3, "Cylinder": 4}
    c = _Cropper(kinds[kind], int(invert), p0, p1, p2, (C.c_double * 3)(*[float(v) for v in centre]))
    return c


def crop_mask(cropper, pts):
    p = np.ascontiguousarray(pts, np.float64)
    m = np.zeros(p.shape[0], np.uint8)
    lib().orc_crop_mask(C.byref(cropper), _d(p), p.shape[0], m.ctypes.data_as(C.POINTER(C.c_uint8)))
    return m.astype(bool)


def voxelize_within_crop(cropper, voxel_size, pts, normals=None):
    p = np.ascontiguousarray(pts, np.float64)
    n = None if normals is None else np.ascontiguousarray(normals, np.float64)
    op = np.zeros_like(p)
    on = np.zeros_like(p)
    oi = np.zeros((p.shape[0], 3), np.int32)
    k = lib().orc_voxelize_within_crop(C.byref(cropper), float(voxel_size), _d(p), _d(n), p.shape[0], _d(op), _d(on), _i(oi))
    return op[:k].copy(), (on[:k].copy() if n is not None else None), oi[:k].copy()


def voxel_downsample_o3d(voxel_size, pts, normals=None):
    p = np.ascontiguousarray(pts, np.float64)
    n = None if normals is None else np.ascontiguousarray(normals, np.float64)
    op = np.zeros_like(p)
    on = np.zeros_like(p)
    oi = np.zeros((p.shape[0], 3), np.int32)
    k = lib().orc_voxel_downsample_o3d(float(voxel_size), _d(p), _d(n), p.shape[0], _d(op), _d(on), _i(oi))
    return op[:k].copy(), (on[:k].copy() if n is not None else None), oi[:k].copy()


def transform_cloud(T, pts, normals=None):
    """o3d_slam::transform (helpers.cpp:283-318), including its doubled output for an (almost-)identity T."""
    p = np.ascontiguousarray(pts, np.float64)
    nn = None if normals is None else np.ascontiguousarray(normals, np.float64)
    Tc = np.ascontiguousarray(np.asarray(T, np.float64).T).reshape(16)
    out = np.zeros((2 * p.shape[0], 3), np.float64)
    outn = None if nn is None else np.zeros_like(out)
    n = lib().orc_transform_cloud(_d(Tc), _d(p), _d(nn), p.shape[0], _d(out), _d(outn))
    return out[:n].copy(), (None if outn is None else outn[:n].copy())


def estimate_normals(pts, radius, max_nn, want_neighbours=False):
    """EstimateNormals(Hybrid(radius, max_nn)) + NormalizeNormals + OrientNormalsTowardsCameraLocation(0); brute force."""
    p = np.ascontiguousarray(pts, np.float64)
    out = np.zeros_like(p)
    nn = np.zeros((p.shape[0], max_nn), np.int32) if want_neighbours else None
    lib().orc_estimate_normals(_d(p), p.shape[0], float(radius), int(max_nn), _d(out), _i(nn))
    return (out, nn) if want_neighbours else out


def o3d_registration_icp(source, target, target_normals, max_correspondence_distance, init=None, relative_fitness=1e-6,
                         relative_rmse=1e-6, max_iteration=30):
    """open3d::pipelines::registration::RegistrationICP(..., TransformationEstimationPointToPlane(), criteria)."""
    s_ = np.ascontiguousarray(source, np.float64)
    t_ = np.ascontiguousarray(target, np.float64)
    n_ = np.ascontiguousarray(target_normals, np.float64)
    T0 = np.ascontiguousarray(np.asarray(np.eye(4) if init is None else init, np.float64).T).reshape(16)
    r = _O3dIcpResult()
    code = lib().orc_o3d_registration_icp(_d(s_), s_.shape[0], _d(t_), _d(n_), t_.shape[0], float(max_correspondence_distance), _d(T0),
                                          float(relative_fitness), float(relative_rmse), int(max_iteration), C.byref(r))
    if code != OK:
        raise OracleError(code)
    return {"transformation": np.array(r.transformation).reshape(4, 4).T.copy(), "fitness": r.fitness, "inlier_rmse": r.inlier_rmse,
            "correspondences": int(r.correspondences), "iterations": int(r.iterations)}


def o3d_information_matrix(source, target, max_correspondence_distance, T):
    s_ = np.ascontiguousarray(source, np.float64)
    t_ = np.ascontiguousarray(target, np.float64)
    Tc = np.ascontiguousarray(np.asarray(T, np.float64).T).reshape(16)
    out = np.zeros(36)
    lib().orc_o3d_information_matrix(_d(s_), s_.shape[0], _d(t_), t_.shape[0], float(max_correspondence_distance), _d(Tc), _d(out))
    return out.reshape(6, 6).T.copy()


def carve(scan_map_frame, map_pts, map_normals, sensor, voxel_size=0.1, max_length=20.0, truncation=0.1, min_dot=0.5, subset=None):
    """getIdxsOfCarvedPoints (helpers.cpp:245-281): boolean mask over the map points, True = carved."""
    sc = np.ascontiguousarray(scan_map_frame, np.float64)
    mp = np.ascontiguousarray(map_pts, np.float64)
    mn = None if map_normals is None else np.ascontiguousarray(map_normals, np.float64)
    sub = None if subset is None else np.ascontiguousarray(subset, np.uint8)
    out = np.zeros(mp.shape[0], np.uint8)
    u8 = C.POINTER(C.c_uint8)
    lib().orc_carve(_d(sc), sc.shape[0], _d(mp), _d(mn), mp.shape[0], None if sub is None else sub.ctypes.data_as(u8),
                    _d(np.ascontiguousarray(sensor, np.float64)), float(voxel_size), float(max_length), float(truncation), float(min_dot),
                    out.ctypes.data_as(u8))
    return out.astype(bool)


class DenseMap:
    """VoxelizedPointCloud (O3S/src/Voxel.cpp:38-114) + Submap::carve on it (O3S/src/Submap.cpp:146-157)."""

    def __init__(self, voxel_size: float):
        self.voxel_size = float(voxel_size)
        self._h = lib().orc_dense_create(self.voxel_size)
        self.has_normals = False

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_dense_destroy(self._h)
            self._h = None

    def size(self) -> int:
        return int(lib().orc_dense_size(self._h))

    def insert(self, pts, normals=None):
        p = np.ascontiguousarray(pts, np.float64)
        n = None if normals is None else np.ascontiguousarray(normals, np.float64)
        lib().orc_dense_insert(self._h, _d(p), _d(n), p.shape[0])
        if n is not None and p.shape[0]:
            self.has_normals = True

    def to_point_cloud(self):
        """(points, normals | None, keys, counts), voxels in ascending (z, y, x) key order."""
        V = self.size()
        pts = np.zeros((V, 3))
        nrm = np.zeros((V, 3))
        keys = np.zeros((V, 3), np.int32)
        cnt = np.zeros(V, np.int32)
        n = int(lib().orc_dense_to_point_cloud(self._h, _d(pts), _d(nrm), _i(keys), _i(cnt)))
        return pts[:n], (nrm[:n] if self.has_normals else None), keys[:n], cnt[:n]

    def transform(self, T):
        Tc = np.ascontiguousarray(np.asarray(T, np.float64).T).reshape(16)
        lib().orc_dense_transform(self._h, _d(Tc))

    def carve(self, scan, sensor_position, neighborhood_radius=0.1, max_length=20.0, truncation=0.1) -> int:
        sc = np.ascontiguousarray(scan, np.float64)
        return int(lib().orc_dense_carve(self._h, _d(sc), sc.shape[0], _d(np.ascontiguousarray(sensor_position, np.float64)),
                                         float(neighborhood_radius), float(max_length), float(truncation)))


def remove_duplicate_points(pts, voxel_size):
    """removeDuplicatePointsWithinSameVoxels (O3S/src/Voxel.cpp:162-192): boolean keep mask (first point of every voxel)."""
    p = np.ascontiguousarray(pts, np.float64)
    keep = np.zeros(p.shape[0], np.uint8)
    lib().orc_remove_duplicate_points(_d(p), p.shape[0], float(voxel_size), keep.ctypes.data_as(C.POINTER(C.c_uint8)))
    return keep.astype(bool)


def voxels_within_neighborhood(p, radius, voxel_size):
    """getVoxelsWithinPointNeighborhood (O3S/src/VoxelHashMap.cpp:13-46): keys in the reference's order, duplicates kept."""
    pp = np.ascontiguousarray(p, np.float64).reshape(3)
    n = int(lib().orc_voxels_within_neighborhood(_d(pp), float(radius), float(voxel_size), None, 0))
    keys = np.zeros((n, 3), np.int32)
    lib().orc_voxels_within_neighborhood(_d(pp), float(radius), float(voxel_size), _i(keys), n)
    return keys


def voxelize_attrs(mode, cropper, voxel_size, pts, colors=None, covariances=None):
    """Colours / covariances of voxelize_within_crop (mode 0) / voxel_downsample_o3d (mode 1), in the order of their points."""
    p = np.ascontiguousarray(pts, np.float64)
    col = None if colors is None else np.ascontiguousarray(colors, np.float64)
    cov = None if covariances is None else np.ascontiguousarray(covariances, np.float64).reshape(-1, 9)
    oc = np.zeros((p.shape[0], 3))
    ov = np.zeros((p.shape[0], 9))
    L = lib()
    L.orc_voxelize_attrs.restype = C.c_int64
    dp = C.POINTER(C.c_double)
    L.orc_voxelize_attrs.argtypes = [C.c_int, C.POINTER(_Cropper), C.c_double, dp, dp, dp, C.c_int64, dp, dp]
    cr = cropper if cropper is not None else make_cropper()
    k = L.orc_voxelize_attrs(int(mode), C.byref(cr), float(voxel_size), _d(p), _d(col), _d(cov), p.shape[0], _d(oc), _d(ov))
    return (oc[:k].copy() if col is not None else None), (ov[:k].copy() if cov is not None else None)


def transform_cov(T, covariances):
    cov = np.ascontiguousarray(covariances, np.float64).reshape(-1, 9)
    Tc = np.ascontiguousarray(np.asarray(T, np.float64).T).reshape(16)
    out = np.zeros((2 * cov.shape[0], 9))
    L = lib()
    L.orc_transform_cov.restype = C.c_int64
    dp = C.POINTER(C.c_double)
    L.orc_transform_cov.argtypes = [dp, dp, C.c_int64, dp]
    n = L.orc_transform_cov(_d(Tc), _d(cov), cov.shape[0], _d(out))
    return out[:n].copy()


def overlap_indices(source, target, T, voxel_size, min_points_per_voxel=1):
    """computeIndicesOfOverlappingPoints (helpers.cpp:319-345); indices ascending."""
    sp = np.ascontiguousarray(source, np.float64)
    tp = np.ascontiguousarray(target, np.float64)
    Tc = np.ascontiguousarray(np.asarray(T, np.float64).T).reshape(16)
    i_s = np.zeros(max(sp.shape[0], 1), np.int64)
    i_t = np.zeros(max(tp.shape[0], 1), np.int64)
    ns, nt = C.c_int64(), C.c_int64()
    L = lib()
    L.orc_overlap_indices.restype = None
    L.orc_overlap_indices.argtypes = [C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double), C.c_double,
                                      C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.orc_overlap_indices(_d(sp), sp.shape[0], _d(tp), tp.shape[0], _d(Tc), float(voxel_size), int(min_points_per_voxel),
                          i_s.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(ns), i_t.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(nt))
    return i_s[:ns.value].copy(), i_t[:nt.value].copy()


def o3d_to_pm(pts, normals=None):
    p = np.ascontiguousarray(pts, np.float64)
    n = None if normals is None else np.ascontiguousarray(normals, np.float64)
    xyzw = np.zeros((p.shape[0], 4), np.float32)
    on = np.zeros((p.shape[0], 3), np.float32) if n is not None else None
    lib().orc_o3d_to_pm(_d(p), _d(n), p.shape[0], _f(xyzw), _f(on))
    return xyzw, on


def pose_error(Ta, Tb):
    """computeError (LPM/testing/utils_transformations.cpp:7-25): delta = Ta^-1 Tb -> (|dt| components, angle)."""
    Ta = np.asarray(Ta, np.float64)
    Tb = np.asarray(Tb, np.float64)
    D = np.linalg.inv(Ta) @ Tb
    dt = D[:3, 3]
    c = (np.trace(D[:3, :3]) - 1.0) / 2.0
    ang = math.acos(max(-1.0, min(1.0, c)))
    # small-angle accurate form
    s = np.linalg.norm([D[2, 1] - D[1, 2], D[0, 2] - D[2, 0], D[1, 0] - D[0, 1]]) / 2.0
    ang = math.atan2(s, c)
    return dt, abs(ang)
