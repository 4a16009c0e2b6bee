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


@dataclass
class RegistrationResult:
    transformation: np.ndarray
    fitness: float
    inlier_rmse: float
    correspondences: int
    iterations: int


_bound = False


def _L():
    global _bound
    L = _lib.lib()
    if not _bound:
        dp = C.POINTER(C.c_double)
        L.o3s_o3d_registration_icp.argtypes = [C.c_int, dp, C.c_int64, dp, dp, C.c_int64, C.c_double, dp, C.POINTER(_Criteria), C.POINTER(_Result)]
        L.o3s_o3d_information_matrix.argtypes = [C.c_int, dp, C.c_int64, dp, C.c_int64, C.c_double, dp, dp]
        _bound = True
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
