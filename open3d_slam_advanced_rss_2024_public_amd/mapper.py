"""Python mirror of cpp/o3s_mapper.hpp (the caller glue of o3d_slam::Mapper::addRangeMeasurement, open3d_slam/src/Mapper.cpp:
168-504, over device-resident scans / submaps / the ICP handle): used by the tests to check the compiled header step by
step.  No compute here — every cloud operation is a call into the C-ABI library; the CPU tests of the control flow pass
stand-ins for the ICP, the submap collection and the cropping volumes."""
import numpy as np


def mul4(A, B):
    """4x4 product in the driver's operation order (plain k = 0..3 accumulation, no FMA)."""
    C_ = np.zeros((4, 4))
    for c in range(4):
        for r in range(4):
            s = A[r, 0] * B[0, c]
            s = s + A[r, 1] * B[1, c]
            s = s + A[r, 2] * B[2, c]
            s = s + A[r, 3] * B[3, c]
            C_[r, c] = s
    return C_


def inv_iso(T):
    R = np.eye(4)
    R[:3, :3] = T[:3, :3].T
    for r in range(3):
        s = R[r, 0] * T[0, 3]
        s = s + R[r, 1] * T[1, 3]
        s = s + R[r, 2] * T[2, 3]
        R[r, 3] = -s
    return R


class Mapper:
    """Mapper::addRangeMeasurement restated over the Python mirror (the same steps as cpp/o3s_mapper.hpp).

    icp / collection: the ICP handle and the SubmapCollection (device-backed in the GPU tests, stand-ins in the CPU tests of
    the control flow); wide / narrow: the map-builder and scan-matcher cropping volumes as objects the scan's preprocess and
    the submap's set_reference accept."""

    def __init__(self, icp, collection, wide, narrow, scan_voxel, ref_period, min_move):
        self.icp = icp
        self.col = collection
        self.wide, self.narrow, self.scan_voxel, self.ref_period, self.min_move = wide, narrow, scan_voxel, ref_period, min_move
        self.ps = None
        self.odom = {}
        self.T = np.eye(4)
        self.T_prev = np.eye(4)
        self.T_last_insert = np.eye(4)
        self.prior = np.eye(4)
        self.last_stamp = self.last_ref = None
        self.new_value = self.ignore_odom = False
        self.flags = (0, 0, 0)
        self.iters = 0
        self.check = None     # set to a callable(scan inputs, state) to validate a step against the oracle
        self.calib_inv = np.eye(4)          # calibration_.inverse(); identity until set, as in the reference (Mapper.hpp) and MapperHip
        self.is_calibration_set = False     # isCalibrationSet_ (Mapper.cpp:66-85): until it is, every scan is refused (:169-174)
        self.use_initial_map = False

    def set_calibration(self, C_):
        self.calib_inv = inv_iso(np.asarray(C_, np.float64))
        self.is_calibration_set = True

    def _odom(self, stamp):
        """getTransform(t, odomToRangeSensorBuffer_) * calibration_.inverse()   (Mapper.cpp:221-222, 270-273)"""
        return mul4(self.odom[stamp], self.calib_inv)

    @property
    def sm(self):
        return self.col.maps[self.col.active]

    def preprocess(self, sp, sn):
        self.ps.preprocess(self.wide, self.scan_voxel, self.narrow, sp, sn)

    def add(self, sp, sn, stamp):
        inserted = refreset = threw = 0
        self.flags = (0, 0, 0)
        if not self.use_initial_map and not self.is_calibration_set:
            return False
        self.ps = self.col.scan_for_next()
        if len(self.sm) == 0:
            self.T_prev = self.T.copy()
            self.preprocess(sp, sn)
            self.col.insert(self.ps, self.T, stamp)
            self.flags = (1, 0, 0)
            return True
        if self.last_stamp is not None and stamp <= self.last_stamp:
            latest = max(self.odom)
            self.T = mul4(self.T_prev, mul4(inv_iso(self._odom(self.last_stamp)), self._odom(latest)))
            self.T_prev = self.T.copy()
            return True
        est = self.T_prev.copy()
        if stamp in self.odom and self.last_stamp is not None and not self.new_value and not self.ignore_odom:
            est = mul4(self.T_prev, mul4(inv_iso(self._odom(self.last_stamp)), self._odom(stamp)))
        self.ignore_odom = False
        self.prior = est
        self.preprocess(sp, sn)
        prior32 = est.astype(np.float32)
        corrected32 = prior32.copy()
        reset = self.new_value or self.last_ref is None or (stamp - self.last_ref) >= self.ref_period
        state = None
        if not reset and self.sm.patch_count(self.narrow, self.T) == 0:   # cropSubmap on every scan: empty patch -> give up (:328-336)
            return False
        try:
            if reset:
                if self.check:
                    state = self.sm.getMapPointCloud()
                self.sm.set_reference(self.narrow, self.T, self.icp)
                self.last_ref = stamp
                refreset = 1
                self.ref_pose = self.T.copy()
                self.ref_state = state
            self.ps.set_reading(self.icp)
            corrected32 = self.icp.compute_resident(prior32)
            self.iters = self.icp.stats.iterations
            if self.check and reset:
                self.check(self, sp, sn, prior32, corrected32)
        except RuntimeError:
            threw = 1
            corrected32 = prior32.copy()
            self.iters = self.icp.stats.iterations
        corrected = corrected32.astype(np.float64)
        if self.new_value:
            self.T_prev = self.T.copy()
            self.new_value = False
            self.ignore_odom = True
            self.flags = (0, refreset, threw)
            return True
        self.T = corrected
        motion = mul4(inv_iso(self.T_last_insert), self.T)
        moved = np.sqrt(motion[0, 3] * motion[0, 3] + motion[1, 3] * motion[1, 3] + motion[2, 3] * motion[2, 3])
        if not (moved < self.min_move):
            self.col.insert(self.ps, self.T, stamp)
            self.T_last_insert = self.T.copy()
            inserted = 1
        self.last_stamp = stamp
        self.T_prev = self.T.copy()
        self.flags = (inserted, refreset, threw)
        return True
