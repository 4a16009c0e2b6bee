"""Host-side mirror of o3d_slam::Submap's map-building calls over the device-resident submap (include/o3s_submap.h).
Method names follow the reference (open3d_slam/src/Submap.cpp, ScanToMapRegistration.cpp); the map cloud stays in HBM."""
from __future__ import annotations

import os
import ctypes as C

import numpy as np

from . import _lib
from .cloud_ops import CropperC, _d
from .icp import ICP



def _L():
    L = _lib.lib()
    if _lib.needs_binding(L, __name__):  # once per loaded library (product or test-hook build)
        dp = C.POINTER(C.c_double)
        vp = C.c_void_p
        L.o3s_submap_create.argtypes = [C.c_int, C.c_double, C.POINTER(CropperC), C.POINTER(vp)]
        L.o3s_submap_destroy.argtypes = [vp]
        L.o3s_submap_destroy.restype = None
        L.o3s_submap_insert_scan.argtypes = [vp, dp, dp, C.c_int64, dp]
        L.o3s_submap_reserve.argtypes = [vp, C.c_int64]
        L.o3s_submap_size.argtypes = [vp]
        L.o3s_submap_size.restype = C.c_int64
        L.o3s_submap_size_bounds.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.o3s_submap_download.argtypes = [vp, dp, dp]
        L.o3s_submap_center.argtypes = [vp, dp]
        L.o3s_submap_upload.argtypes = [vp, dp, dp, C.c_int64]
        L.o3s_submap_set_reference.argtypes = [vp, C.POINTER(CropperC), dp, vp, C.POINTER(C.c_int64)]
        L.o3s_submap_patch_count.argtypes = [vp, C.POINTER(CropperC), dp, C.POINTER(C.c_int64)]
        L.o3s_submap_insert_processed.argtypes = [vp, vp, dp]
        L.o3s_submap_carve.argtypes = [vp, C.POINTER(CarvingParamsC), dp, C.c_int64, dp, C.POINTER(C.c_int64)]
        L.o3s_scan_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.o3s_scan_destroy.argtypes = [vp]
        L.o3s_scan_destroy.restype = None
        L.o3s_scan_preprocess.argtypes = [vp, C.POINTER(CropperC), C.c_double, C.POINTER(CropperC), dp, dp, C.c_int64,
                                          C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.o3s_scan_get.argtypes = [vp, C.c_int, dp, dp]
        L.o3s_scan_get.restype = C.c_int64
        L.o3s_scan_set_reading.argtypes = [vp, vp]
        L.o3s_scan_set_normal_estimation.argtypes = [vp, C.c_double, C.c_int32]
    return L


def _pose(T) -> np.ndarray:
    """4x4 -> Eigen::Matrix4d::data() order (column-major)."""
    return np.ascontiguousarray(np.asarray(T, np.float64).T).reshape(16)


class CarvingParamsC(C.Structure):
    _fields_ = [("voxel_size", C.c_double), ("max_raytracing_length", C.c_double), ("truncation_distance", C.c_double),
                ("min_dot_product_with_normal", C.c_double)]


class Submap:
    """The active submap's sparse map cloud, resident on one MI355X."""

    def __init__(self, map_voxel_size: float, map_builder_cropper: CropperC, device: int = 0):
        self._lib = _L()   # the library this handle belongs to (product or a hooks build): every later call goes through it
        self._pid = os.getpid()   # _lib.forked_copy: a forked child must not destroy the handle
        self._h = C.c_void_p()
        rc = self._lib.o3s_submap_create(device, float(map_voxel_size), C.byref(map_builder_cropper), C.byref(self._h))
        if rc != _lib.OK:
            self._h = C.c_void_p()
            raise RuntimeError(f"o3s_submap_create failed with o3s_status {rc} (no CPU fallback)")
        self.has_normals = None
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            if not _lib.forked_copy(self):   # a forked child drops its copy of the wrapper, the handle is the parent's
                self._lib.o3s_submap_destroy(self._h)
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
        self._check(self._lib.o3s_submap_insert_scan(self._h, _d(p), _d(n), p.shape[0], _d(_pose(mapToRangeSensor))), "o3s_submap_insert_scan")
        if p.shape[0]:
            self.has_normals = n is not None
        return True

    insert_scan = insertScan

    def insertScanColored(self, points, normals, colors, mapToRangeSensor) -> bool:
        """Submap::insertScan for a coloured scan (o3s_submap_insert_scan_colored)."""
        p = np.ascontiguousarray(points, np.float64)
        n = None if normals is None else np.ascontiguousarray(normals, np.float64)
        c = np.ascontiguousarray(colors, np.float64)
        L = self._lib
        L.o3s_submap_insert_scan_colored.argtypes = None
        self._check(L.o3s_submap_insert_scan_colored(self._h, _d(p), _d(n), _d(c), C.c_int64(p.shape[0]), _d(_pose(mapToRangeSensor))),
                    "o3s_submap_insert_scan_colored")
        if p.shape[0]:
            self.has_normals = n is not None
        return True

    def hasColors(self) -> bool:
        return bool(self._lib.o3s_submap_has_colors(self._h))

    def getMapColors(self):
        out = np.zeros((len(self), 3))
        L = self._lib
        L.o3s_submap_download_colors.argtypes = None
        self._check(L.o3s_submap_download_colors(self._h, _d(out)), "o3s_submap_download_colors")
        return out

    def __len__(self) -> int:
        return int(self._lib.o3s_submap_size(self._h))

    def size_bounds(self):
        """(at_least, at_most) without waiting for an insert whose completion is pending (o3s_submap_size_bounds): equal when none is."""
        lo, hi = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.o3s_submap_size_bounds(self._h, C.byref(lo), C.byref(hi)), "o3s_submap_size_bounds")
        return int(lo.value), int(hi.value)

    def clone(self, device: int = None) -> "Submap":
        """A second submap object with a copy of the map cloud, on the same or another device (o3s_submap_clone): the snapshot a
        loop-closure worker refines while the mapper keeps inserting into the original."""
        L = self._lib
        L.o3s_submap_clone.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        other = object.__new__(Submap)
        other._lib = L
        other._pid = os.getpid()
        other._h = C.c_void_p()
        other.has_normals = self.has_normals
        dev = int(getattr(self, "device", 0) if device is None else device)
        rc = L.o3s_submap_clone(self._h, dev, C.byref(other._h))
        if rc != _lib.OK:
            other._h = C.c_void_p()
            raise RuntimeError(f"o3s_submap_clone failed with o3s_status {rc}")
        other.device = dev
        return other

    def insert_stats(self):
        """(merged, sorted, fell_back): how the voxelising inserts ran (o3s_submap_insert_stats)."""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        L = self._lib
        L.o3s_submap_insert_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        self._check(L.o3s_submap_insert_stats(self._h, C.byref(a), C.byref(b), C.byref(c)), "o3s_submap_insert_stats")
        return int(a.value), int(b.value), int(c.value)

    def reserve(self, n_points: int):
        """Room for n_points (SubmapParameters::maxNumPoints_ + one scan) up front: no re-allocation stall while the map grows."""
        self._check(self._lib.o3s_submap_reserve(self._h, int(n_points)), "o3s_submap_reserve")

    def trim(self):
        """o3s_submap_trim: a submap that is no longer inserted into gives everything but its map cloud back to the allocator."""
        self._lib.o3s_submap_trim.argtypes = [C.c_void_p]
        self._check(self._lib.o3s_submap_trim(self._h), "o3s_submap_trim")

    def hand_over(self, fresh: "Submap"):
        """o3s_submap_hand_over: this (closed) submap keeps its map in arrays of its own size; every other device buffer moves to the
        empty submap `fresh`."""
        self._lib.o3s_submap_hand_over.argtypes = [C.c_void_p, C.c_void_p]
        self._check(self._lib.o3s_submap_hand_over(self._h, fresh._h), "o3s_submap_hand_over")

    def device_bytes(self) -> int:
        self._lib.o3s_submap_device_bytes.argtypes = [C.c_void_p]
        self._lib.o3s_submap_device_bytes.restype = C.c_int64
        return int(self._lib.o3s_submap_device_bytes(self._h))

    def computeSubmapCenter(self) -> np.ndarray:
        """Submap::computeSubmapCenter (Submap.cpp:282-286): open3d GetCenter() of the map cloud, summed on the device."""
        c = np.zeros(3)
        self._check(self._lib.o3s_submap_center(self._h, _d(c)), "o3s_submap_center")
        return c

    def getMapPointCloud(self):
        """(points, normals) copied to the host — for inspection / saving; the ICP never needs it."""
        n = len(self)
        pts = np.zeros((n, 3), np.float64)
        nrm = np.zeros((n, 3), np.float64) if self.has_normals else None
        self._check(self._lib.o3s_submap_download(self._h, _d(pts), _d(nrm)), "o3s_submap_download")
        return pts, nrm

    def setMapPointCloud(self, points, normals):
        p = np.ascontiguousarray(points, np.float64)
        n = None if normals is None else np.ascontiguousarray(normals, np.float64)
        self._check(self._lib.o3s_submap_upload(self._h, _d(p), _d(n), p.shape[0]), "o3s_submap_upload")
        self.has_normals = (n is not None) if p.shape[0] else None

    def carve(self, rawScan, mapToRangeSensor, voxel_size=0.1, max_raytracing_length=20.0, truncation_distance=0.1,
              min_dot_product_with_normal=0.5) -> int:
        """Submap::carve (Submap.cpp:116-130) with SpaceCarvingParameters; returns the number of removed map points.
        The caller applies the cadence (isCarvingEnabled_, carveSpaceEveryNscans_) and calls it BEFORE the insert."""
        p = np.ascontiguousarray(rawScan, np.float64)
        cp = CarvingParamsC(float(voxel_size), float(max_raytracing_length), float(truncation_distance), float(min_dot_product_with_normal))
        k = C.c_int64()
        self._check(self._lib.o3s_submap_carve(self._h, C.byref(cp), _d(p), p.shape[0], _d(_pose(mapToRangeSensor)), C.byref(k)), "o3s_submap_carve")
        return int(k.value)

    def insertProcessed(self, scan: "ProcessedScan", mapToRangeSensor) -> bool:
        """insertScan(rawScan, *processed.merge_, mapToRangeSensor) (Mapper.cpp:487) from the resident merge cloud."""
        self._check(self._lib.o3s_submap_insert_processed(self._h, scan._h, _d(_pose(mapToRangeSensor))), "o3s_submap_insert_processed")
        if scan.n_merge:
            self.has_normals = True
        return True

    def patch_count(self, scan_matcher_cropper: CropperC, mapToRangeSensor) -> int:
        """Size of the patch cropSubmap would return at this pose (Mapper.cpp:328), counted on the device."""
        k = C.c_int64()
        self._check(self._lib.o3s_submap_patch_count(self._h, C.byref(scan_matcher_cropper), _d(_pose(mapToRangeSensor)), C.byref(k)), "o3s_submap_patch_count")
        return int(k.value)

    def set_reference(self, scan_matcher_cropper: CropperC, mapToRangeSensor, icp: ICP) -> int:
        """cropSubmap + open3dToPointmatcher + icp.initReference (Mapper.cpp:328-366) without leaving HBM.
        Returns the patch size; raises if the patch is empty ("Map patch is empty", Mapper.cpp:330-336)."""
        k = C.c_int64()
        rc = self._lib.o3s_submap_set_reference(self._h, C.byref(scan_matcher_cropper), _d(_pose(mapToRangeSensor)), icp._h, C.byref(k))
        if rc == _lib.ERR_EMPTY_REFERENCE:
            raise RuntimeError("map patch is empty")
        if rc != _lib.OK:
            msg = icp._L.o3s_last_error(icp._h).decode()
            raise RuntimeError(f"o3s_submap_set_reference failed with o3s_status {rc}: {msg}")
        return int(k.value)


class ProcessedScan:
    """ScanToMapIcp::processForScanMatchingAndMerging (ScanToMapRegistration.cpp:36-69) with both result clouds resident
    in HBM: ``merge`` (wide crop, voxelised) feeds Submap.insertProcessed, ``match`` (narrow crop) feeds the ICP."""

    def __init__(self, device: int = 0):
        self._lib = _L()   # the library this handle belongs to (product or a hooks build): every later call goes through it
        self._pid = os.getpid()   # _lib.forked_copy: a forked child must not destroy the handle
        self._h = C.c_void_p()
        rc = self._lib.o3s_scan_create(device, C.byref(self._h))
        if rc != _lib.OK:
            self._h = C.c_void_p()
            raise RuntimeError(f"o3s_scan_create failed with o3s_status {rc} (no CPU fallback)")
        self.n_merge = self.n_match = 0

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            if not _lib.forked_copy(self):   # a forked child drops its copy of the wrapper, the handle is the parent's
                self._lib.o3s_scan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_normal_estimation(self, max_radius: float, knn: int):
        """icp.max_distance_knn / icp.knn of the parameter files: used only for scans that arrive without normals."""
        rc = self._lib.o3s_scan_set_normal_estimation(self._h, float(max_radius), int(knn))
        if rc != _lib.OK:
            raise ValueError("knn must be in 1..32 and max_radius > 0")

    def preprocess(self, map_builder_cropper: CropperC, voxel_size: float, scan_matcher_cropper: CropperC, points, normals):
        p = np.ascontiguousarray(points, np.float64)
        n = None if normals is None else np.ascontiguousarray(normals, np.float64)
        a, b = C.c_int64(), C.c_int64()
        rc = self._lib.o3s_scan_preprocess(self._h, C.byref(map_builder_cropper), float(voxel_size), C.byref(scan_matcher_cropper), _d(p), _d(n),
                                      p.shape[0], C.byref(a), C.byref(b))
        if rc == _lib.ERR_BAD_SHAPE:
            raise RuntimeError("the scan has no normals and set_normal_estimation() was not called")
        if rc != _lib.OK:
            raise RuntimeError(f"o3s_scan_preprocess failed with o3s_status {rc}")
        self.n_merge, self.n_match = int(a.value), int(b.value)
        return self.n_merge, self.n_match

    def _get(self, which):
        n = int(self._lib.o3s_scan_get(self._h, which, None, None))
        pts, nrm = np.zeros((n, 3), np.float64), np.zeros((n, 3), np.float64)
        if n and self._lib.o3s_scan_get(self._h, which, _d(pts), _d(nrm)) != n:
            raise RuntimeError("o3s_scan_get failed")
        return pts, nrm

    @property
    def merge(self):
        return self._get(0)

    @property
    def match(self):
        return self._get(1)

    def set_reading(self, icp: ICP):
        """open3dToPointmatcher(*processed.match_) -> resident reading of `icp` (then icp.compute_resident(T_init))."""
        rc = self._lib.o3s_scan_set_reading(self._h, icp._h)
        if rc != _lib.OK:
            raise RuntimeError(f"o3s_scan_set_reading failed with o3s_status {rc}")
