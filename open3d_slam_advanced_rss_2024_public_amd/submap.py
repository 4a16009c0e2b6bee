"""Host-side mirror of o3d_slam::Submap's map-building calls over the device-resident submap (include/o3s_submap.h).
Method names follow the reference (open3d_slam/src/Submap.cpp, ScanToMapRegistration.cpp); the map cloud stays in HBM."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .cloud_ops import CropperC, _d
from .icp import ICP

_bound = False


def _L():
    global _bound
    L = _lib.lib()
    if not _bound:
        dp = C.POINTER(C.c_double)
        vp = C.c_void_p
        L.o3s_submap_create.argtypes = [C.c_int, C.c_double, C.POINTER(CropperC), C.POINTER(vp)]
        L.o3s_submap_destroy.argtypes = [vp]
        L.o3s_submap_destroy.restype = None
        L.o3s_submap_insert_scan.argtypes = [vp, dp, dp, C.c_int64, dp]
        L.o3s_submap_size.argtypes = [vp]
        L.o3s_submap_size.restype = C.c_int64
        L.o3s_submap_download.argtypes = [vp, dp, dp]
        L.o3s_submap_upload.argtypes = [vp, dp, dp, C.c_int64]
        L.o3s_submap_set_reference.argtypes = [vp, C.POINTER(CropperC), dp, vp, C.POINTER(C.c_int64)]
        _bound = True
    return L


def _pose(T) -> np.ndarray:
    """4x4 -> Eigen::Matrix4d::data() order (column-major)."""
    return np.ascontiguousarray(np.asarray(T, np.float64).T).reshape(16)


class Submap:
    """The active submap's sparse map cloud, resident on one MI355X."""

    def __init__(self, map_voxel_size: float, map_builder_cropper: CropperC, device: int = 0):
        self._h = C.c_void_p()
        rc = _L().o3s_submap_create(device, float(map_voxel_size), C.byref(map_builder_cropper), C.byref(self._h))
        if rc != _lib.OK:
            self._h = C.c_void_p()
            raise RuntimeError(f"o3s_submap_create failed with o3s_status {rc} (no CPU fallback)")
        self.has_normals = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            _L().o3s_submap_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != _lib.OK:
            raise RuntimeError(f"{what} failed with o3s_status {rc}")

    def insertScan(self, points, normals, mapToRangeSensor) -> bool:
        """Submap::insertScan (Submap.cpp:39-96) without carving."""
        p = np.ascontiguousarray(points, np.float64)
        n = None if normals is None else np.ascontiguousarray(normals, np.float64)
        self._check(_L().o3s_submap_insert_scan(self._h, _d(p), _d(n), p.shape[0], _d(_pose(mapToRangeSensor))), "o3s_submap_insert_scan")
        if p.shape[0]:
            self.has_normals = n is not None
        return True

    insert_scan = insertScan

    def __len__(self) -> int:
        return int(_L().o3s_submap_size(self._h))

    def getMapPointCloud(self):
        """(points, normals) copied to the host — for inspection / saving; the ICP never needs it."""
        n = len(self)
        pts = np.zeros((n, 3), np.float64)
        nrm = np.zeros((n, 3), np.float64) if self.has_normals else None
        self._check(_L().o3s_submap_download(self._h, _d(pts), _d(nrm)), "o3s_submap_download")
        return pts, nrm

    def setMapPointCloud(self, points, normals):
        p = np.ascontiguousarray(points, np.float64)
        n = None if normals is None else np.ascontiguousarray(normals, np.float64)
        self._check(_L().o3s_submap_upload(self._h, _d(p), _d(n), p.shape[0]), "o3s_submap_upload")
        self.has_normals = (n is not None) if p.shape[0] else None

    def set_reference(self, scan_matcher_cropper: CropperC, mapToRangeSensor, icp: ICP) -> int:
        """cropSubmap + open3dToPointmatcher + icp.initReference (Mapper.cpp:328-366) without leaving HBM.
        Returns the patch size; raises if the patch is empty ("Map patch is empty", Mapper.cpp:330-336)."""
        k = C.c_int64()
        rc = _L().o3s_submap_set_reference(self._h, C.byref(scan_matcher_cropper), _d(_pose(mapToRangeSensor)), icp._h, C.byref(k))
        if rc == _lib.ERR_EMPTY_REFERENCE:
            raise RuntimeError("map patch is empty")
        if rc != _lib.OK:
            msg = icp._L.o3s_last_error(icp._h).decode()
            raise RuntimeError(f"o3s_submap_set_reference failed with o3s_status {rc}: {msg}")
        return int(k.value)
