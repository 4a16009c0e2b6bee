"""Host-side mirror of o3d_slam::VoxelizedPointCloud and of Submap's dense-map calls over the device-resident dense map
(include/o3s_dense_map.h).  Method names follow the reference (open3d_slam/src/Voxel.cpp:38-114,
open3d_slam/src/Submap.cpp:97-113,146-157); the voxel table stays in HBM."""
from __future__ import annotations

import os
import ctypes as C

import numpy as np

from . import _lib
from .cloud_ops import CropperC, _d



class DenseCarvingParamsC(C.Structure):
    """The dense-map fields of SpaceCarvingParameters (open3d_slam/include/open3d_slam/Parameters.hpp:88-95)."""
    _fields_ = [("neighborhood_radius_dense_map", C.c_double), ("max_raytracing_length", C.c_double),
                ("truncation_distance", C.c_double), ("carve_space_every_n_scans", C.c_int32), ("reserved", C.c_int32)]

    @classmethod
    def make(cls, neighborhood_radius=0.1, max_raytracing_length=20.0, truncation_distance=0.1, carve_space_every_n_scans=10):
        return cls(float(neighborhood_radius), float(max_raytracing_length), float(truncation_distance), int(carve_space_every_n_scans), 0)


def _L():
    L = _lib.lib()
    if _lib.needs_binding(L, __name__):  # once per loaded library (product or test-hook build)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        vp = C.c_void_p
        i64p = C.POINTER(C.c_int64)
        L.o3s_dense_map_create.argtypes = [C.c_int, C.c_double, C.POINTER(vp)]
        L.o3s_dense_map_destroy.argtypes = [vp]
        L.o3s_dense_map_destroy.restype = None
        L.o3s_dense_map_clear.argtypes = [vp]
        L.o3s_dense_map_clear.restype = None
        L.o3s_dense_map_size.argtypes = [vp]
        L.o3s_dense_map_size.restype = C.c_int64
        L.o3s_dense_map_has_normals.argtypes = [vp]
        L.o3s_dense_map_insert.argtypes = [vp, dp, dp, C.c_int64]
        L.o3s_dense_map_insert_scan.argtypes = [vp, C.POINTER(CropperC), dp, dp, C.c_int64, dp, C.POINTER(DenseCarvingParamsC), i64p]
        L.o3s_dense_map_insert_resident_scan.argtypes = [vp, C.POINTER(CropperC), vp, dp, C.POINTER(DenseCarvingParamsC), i64p]
        L.o3s_dense_map_carve.argtypes = [vp, C.POINTER(DenseCarvingParamsC), dp, C.c_int64, dp, i64p]
        L.o3s_dense_map_to_point_cloud.argtypes = [vp, dp, dp, ip, ip, i64p]
        L.o3s_dense_map_transform.argtypes = [vp, dp]
    return L


def _pose(T) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(T, np.float64).T).reshape(16)


class DenseMap:
    """VoxelizedPointCloud resident on one MI355X."""

    def __init__(self, voxel_size: float, device: int = 0):
        self._lib = _L()   # the library this handle belongs to (product or a hooks build): every later call goes through it
        self._pid = os.getpid()   # _lib.forked_copy: a forked child must not destroy the handle
        self._h = C.c_void_p()
        self.voxel_size = float(voxel_size)
        rc = self._lib.o3s_dense_map_create(device, self.voxel_size, C.byref(self._h))
        if rc != _lib.OK:
            self._h = C.c_void_p()
            raise RuntimeError(f"o3s_dense_map_create failed with o3s_status {rc} (no CPU fallback)")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            if not _lib.forked_copy(self):   # a forked child drops its copy of the wrapper, the handle is the parent's
                self._lib.o3s_dense_map_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _check(rc, what):
        if rc != _lib.OK:
            raise RuntimeError(f"{what} failed with o3s_status {rc}")

    def size(self) -> int:
        return int(self._lib.o3s_dense_map_size(self._h))

    def empty(self) -> bool:
        return self.size() == 0

    def hasNormals(self) -> bool:
        return bool(self._lib.o3s_dense_map_has_normals(self._h))

    def clear(self):
        self._lib.o3s_dense_map_clear(self._h)

    def insert(self, points, normals=None):
        """VoxelizedPointCloud::insert: a cloud already in the map frame."""
        p = np.ascontiguousarray(points, np.float64)
        n = None if normals is None else np.ascontiguousarray(normals, np.float64)
        self._check(self._lib.o3s_dense_map_insert(self._h, _d(p), _d(n), p.shape[0]), "o3s_dense_map_insert")

    def insertScanDenseMap(self, raw_points, T_map_sensor, dense_map_cropper: CropperC, raw_normals=None, carving: DenseCarvingParamsC | None = None) -> int:
        """Submap::insertScanDenseMap; carving=None means isPerformCarving == false.  Returns the voxels carved away."""
        p = np.ascontiguousarray(raw_points, np.float64)
        n = None if raw_normals is None else np.ascontiguousarray(raw_normals, np.float64)
        removed = C.c_int64(0)
        self._check(self._lib.o3s_dense_map_insert_scan(self._h, C.byref(dense_map_cropper), _d(p), _d(n), p.shape[0], _d(_pose(T_map_sensor)),
                                                   None if carving is None else C.byref(carving), C.byref(removed)), "o3s_dense_map_insert_scan")
        return int(removed.value)

    def insertResidentScanDenseMap(self, processed_scan, T_map_sensor, dense_map_cropper: CropperC, carving: DenseCarvingParamsC | None = None) -> int:
        """Submap::insertScanDenseMap with the raw scan that ProcessedScan.preprocess left in HBM (no second upload)."""
        removed = C.c_int64(0)
        self._check(self._lib.o3s_dense_map_insert_resident_scan(self._h, C.byref(dense_map_cropper), processed_scan._h, _d(_pose(T_map_sensor)),
                                                            None if carving is None else C.byref(carving), C.byref(removed)),
                    "o3s_dense_map_insert_resident_scan")
        return int(removed.value)

    def carve(self, scan_points, sensor_position, carving: DenseCarvingParamsC) -> int:
        """Submap::carve(scan, sensorPosition, param, &denseMap_) without the every-N-scans gate."""
        p = np.ascontiguousarray(scan_points, np.float64)
        removed = C.c_int64(0)
        self._check(self._lib.o3s_dense_map_carve(self._h, C.byref(carving), _d(p), p.shape[0], _d(np.ascontiguousarray(sensor_position, np.float64)),
                                             C.byref(removed)), "o3s_dense_map_carve")
        return int(removed.value)

    def toPointCloud(self, with_keys: bool = False):
        """(points, normals | None) — plus (keys, counts) with with_keys — in ascending (z, y, x) voxel order."""
        V = self.size()
        pts = np.zeros((V, 3))
        nrm = np.zeros((V, 3)) if self.hasNormals() else None
        keys = np.zeros((V, 3), np.int32)
        cnt = np.zeros(V, np.int32)
        n = C.c_int64(0)
        self._check(self._lib.o3s_dense_map_to_point_cloud(self._h, _d(pts), _d(nrm), keys.ctypes.data_as(C.POINTER(C.c_int32)),
                                                      cnt.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n)), "o3s_dense_map_to_point_cloud")
        k = int(n.value)
        out = (pts[:k], None if nrm is None else nrm[:k])
        return out + (keys[:k], cnt[:k]) if with_keys else out

    def transform(self, T):
        self._check(self._lib.o3s_dense_map_transform(self._h, _d(_pose(T))), "o3s_dense_map_transform")
