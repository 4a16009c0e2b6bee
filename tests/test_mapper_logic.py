"""Host logic of Mapper::addRangeMeasurement (open3d_slam/src/Mapper.cpp:168-504) on the CPU: the Python mirror of
cpp/o3s_mapper.hpp (open3d_slam_advanced_rss_2024_public_amd/mapper.py) driven with stand-ins — an "ICP" that returns a
scripted correction of the prior or throws, submaps that only count what they are given — against expectations traced by hand
from the reference's lines.  All poses are pure translations along x, so every product is a sum that can be checked on
paper.  The compiled header is checked against the same mirror, with real device objects, in tests/test_gpu_mapper_cpp.py."""
import numpy as np

from open3d_slam_advanced_rss_2024_public_amd.mapper import Mapper
from open3d_slam_advanced_rss_2024_public_amd.submap_collection import SubmapCollection


class FakeScan:
    def __init__(self):
        self.n = 0

    def preprocess(self, wide, voxel, narrow, pts, normals):
        self.n = len(pts)
        return self.n, self.n

    def set_reading(self, icp):
        if self.n == 0:
            raise RuntimeError("narrow cropped size is zero")      # ScanToMapRegistration.cpp:66 -> libpointmatcher throws
        icp.reading = self


class FakeSubmap:
    def __init__(self):
        self.inserted, self.ref_poses = [], []

    def insertProcessed(self, scan, T):
        self.inserted.append(float(np.asarray(T)[0, 3]))
        return True

    def __len__(self):
        return len(self.inserted)

    def set_reference(self, cropper, T, icp):
        self.ref_poses.append(float(np.asarray(T)[0, 3]))
        return 1

    def patch_count(self, cropper, T):
        self.patch_checks = getattr(self, "patch_checks", 0) + 1
        return getattr(self, "patch_size", 1)

    def computeSubmapCenter(self):
        return np.array([np.mean(self.inserted), 0.0, 0.0])


class Stats:
    iterations = 3


class FakeIcp:
    """compute_resident(prior) = prior shifted by `dx` along x (in float32, like the PmTfParameters the reference gets back)."""

    def __init__(self, dx=0.25):
        self.dx, self.stats, self.calls = dx, Stats(), []

    def compute_resident(self, prior32):
        self.calls.append(float(prior32[0, 3]))
        T = np.array(prior32, np.float32)
        T[0, 3] += np.float32(self.dx)
        return T


def tx(x):
    T = np.eye(4)
    T[0, 3] = x
    return T


def make(ref_period=0.25, min_move=0.0, dx=0.25, calibration=None):
    col = SubmapCollection(1e12, 5, 10 ** 12, 3, 0.1, ("MaxRadius", 30.0), submap_factory=FakeSubmap, scan_factory=FakeScan)
    m = Mapper(FakeIcp(dx), col, "wide", "narrow", 0.1, ref_period, min_move)
    m.set_calibration(np.eye(4) if calibration is None else calibration)
    return m


PTS = np.zeros((10, 3))


def test_first_scan_is_inserted_at_the_given_pose_without_registration():
    m = make()
    m.T = tx(5.0)                                    # setMapToRangeSensor before the first scan
    assert m.add(PTS, PTS, 0.0)
    assert m.flags == (1, 0, 0) and m.icp.calls == []      # Mapper.cpp:179-195: no ICP, no reference
    assert m.sm.inserted == [5.0] and m.T[0, 3] == 5.0 and m.T_prev[0, 3] == 5.0


def test_prior_is_previous_pose_times_odometry_motion_and_the_reference_is_renewed_by_period():
    """Scans every 0.1 s, odometry x = 100 + 2 k (its own frame), the ICP adds 0.25 to whatever prior it gets, reference renewed
    every 0.25 s.  By hand: scan 1 has no previous stamp (lastMeasurementTimestamp_ is still unset after the first scan), so its
    prior is the previous pose 5.0 -> 5.25; from then on prior_k = T_{k-1} + 2 (Mapper.cpp:265-281): 7.25 -> 7.5, 9.5 -> 9.75, ...
    The reference is (re)initialised at stamps 0.1, 0.4, 0.7 (first time, then whenever >= 0.25 s have passed: :349)."""
    m = make(ref_period=0.25)
    m.T = tx(5.0)
    flags = []
    for k in range(8):
        m.odom[round(0.1 * k, 10)] = tx(100.0 + 2.0 * k)
        assert m.add(PTS, PTS, round(0.1 * k, 10))
        flags.append(m.flags)
    assert m.icp.calls == [5.0, 7.25, 9.5, 11.75, 14.0, 16.25, 18.5]
    assert m.T[0, 3] == 18.75
    assert [f[1] for f in flags] == [0, 1, 0, 0, 1, 0, 0, 1]
    # poses after scans 0..7: 5, 5.25, 7.5, 9.75, 12, 14.25, 16.5, 18.75; cropSubmap uses mapToRangeSensor_ = the pose BEFORE
    # this scan's result (:328), i.e. the poses after scans 0, 3 and 6
    assert m.sm.ref_poses == [5.0, 9.75, 16.5]
    assert [f[0] for f in flags] == [1] * 8              # minMovementBetweenMappingSteps_ = 0: every scan is merged


def test_minimum_movement_gates_the_insert():
    m = make(min_move=0.6, dx=0.25)
    m.T = tx(0.0)
    ins = []
    for k in range(6):
        assert m.add(PTS, PTS, 0.1 * k)             # no odometry: prior = previous pose, the ICP adds 0.25 each time
        ins.append(m.flags[0])
    # poses 0, 0.25, 0.5, 0.75, 1.0, 1.25; mapToRangeSensorLastScanInsertion_ starts at the identity = 0 (never set by the
    # first scan: Mapper.cpp:483-489 compares against it all the same): inserts at 0 (first scan), 0.75 (moved 0.75 from 0), ...
    assert ins == [1, 0, 0, 1, 0, 0] and m.sm.inserted == [0.0, 0.75]


def test_an_icp_exception_keeps_the_prior_through_the_float_cast():
    m = make()
    m.T = tx(1.0 / 3.0)
    assert m.add(PTS, PTS, 0.0)
    m.odom[0.0] = tx(0.0)
    m.odom[0.1] = tx(0.0)
    assert m.add(np.zeros((0, 3)), np.zeros((0, 3)), 0.1)          # empty reading: set_reading throws
    assert m.flags[2] == 1 and m.icp.calls == []
    # Mapper.cpp:420-435: mapToRangeSensorEstimate stays, but it has been through PmTfParameters (float) and back (double)
    assert m.T[0, 3] == float(np.float32(1.0 / 3.0)) and m.T[0, 3] != 1.0 / 3.0


def test_pose_reset_adopts_the_given_pose_and_skips_odometry_once():
    m = make()
    m.T = tx(0.0)
    for k in range(3):
        m.odom[round(0.1 * k, 10)] = tx(10.0 * k)
        assert m.add(PTS, PTS, round(0.1 * k, 10))
    # setMapToRangeSensorInitial(50): Mapper.cpp:96-118
    m.T = tx(50.0)
    m.T_prev = tx(50.0)
    m.new_value = True
    m.odom[0.3] = tx(30.0)
    assert m.add(PTS, PTS, 0.3)
    assert m.flags == (0, 1, 0)                     # reference renewed at the new pose, the scan's own result and insert are skipped (:440-455)
    assert m.icp.calls[-1] == 50.0 and m.T[0, 3] == 50.0 and m.sm.ref_poses[-1] == 50.0
    m.odom[0.4] = tx(40.0)
    assert m.add(PTS, PTS, 0.4)
    assert m.icp.calls[-1] == 50.0                  # isIgnoreOdometryPrediction_: no odometry motion on the scan after the reset
    m.odom[0.5] = tx(50.0)
    assert m.add(PTS, PTS, 0.5)
    # ... and lastMeasurementTimestamp_ was left at 0.2 by the reset scan (:440-455 returns early) but updated at 0.4: motion 10
    assert m.icp.calls[-1] == 50.25 + 10.0


def test_out_of_order_stamp_propagates_by_odometry_only():
    m = make()
    m.T = tx(0.0)
    for k in range(3):
        m.odom[round(0.1 * k, 10)] = tx(10.0 * k)
        assert m.add(PTS, PTS, round(0.1 * k, 10))
    pose, calls = m.T[0, 3], len(m.icp.calls)
    m.odom[0.25] = tx(27.0)                          # the latest odometry pose
    assert m.add(PTS, PTS, 0.15)                     # stamp <= lastMeasurementTimestamp_ (0.2): Mapper.cpp:197-235
    assert len(m.icp.calls) == calls and m.flags == (0, 0, 0)
    assert m.T[0, 3] == pose + (27.0 - 20.0)         # previous pose * (odom(last stamp)^-1 * odom(latest))


def test_no_scan_is_accepted_before_the_calibration_is_set():
    """Mapper.cpp:169-174: "Calibration is not set. Returning from mapping." — unless the mapper runs on an initial map."""
    col = SubmapCollection(1e12, 5, 10 ** 12, 3, 0.1, ("MaxRadius", 30.0), submap_factory=FakeSubmap, scan_factory=FakeScan)
    m = Mapper(FakeIcp(), col, "wide", "narrow", 0.1, 0.25, 0.0)
    assert not m.add(PTS, PTS, 0.0) and len(m.sm) == 0
    m.set_calibration(np.eye(4))
    assert m.add(PTS, PTS, 0.0) and len(m.sm) == 1


def test_calibration_is_taken_off_every_odometry_pose_before_the_motion_is_formed():
    """Mapper.cpp:270-281: odomToRangeSensor = odom(t) * C^-1, motion = prev^-1 * now = C * odom(prev)^-1 * odom(now) * C^-1.
    With C a quarter turn about z and odometry steps of +2 along the odometry frame's x, the sensor-frame motion is +2 along the
    SENSOR's x = R_C applied to (2, 0, 0) = (0, 2, 0): the prior moves along y, not x (by hand: C = Rz(90) => C (2,0,0) = (0,2,0))."""
    Cq = np.eye(4)
    Cq[:3, :3] = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    m = make(dx=0.0, calibration=Cq)
    m.T = tx(5.0)
    for k in range(3):
        m.odom[round(0.1 * k, 10)] = tx(100.0 + 2.0 * k)
        assert m.add(PTS, PTS, round(0.1 * k, 10))
    # scan 1: no previous stamp yet -> prior = previous pose; scan 2: previous pose * motion, motion = translation (0, 2, 0)
    assert np.allclose(m.prior[:3, 3], [5.0, 2.0, 0.0], atol=1e-12) and np.allclose(m.prior[:3, :3], np.eye(3), atol=1e-12)
    # the identity calibration moves it along x instead
    m2 = make(dx=0.0)
    m2.T = tx(5.0)
    for k in range(3):
        m2.odom[round(0.1 * k, 10)] = tx(100.0 + 2.0 * k)
        assert m2.add(PTS, PTS, round(0.1 * k, 10))
    assert np.allclose(m2.prior[:3, 3], [7.0, 0.0, 0.0], atol=1e-12)


def test_an_empty_map_patch_gives_the_scan_up_between_two_reference_renewals_too():
    """Mapper.cpp:328-336: cropSubmap runs on EVERY scan and an empty patch returns false — also on the scans that do not renew
    the ICP reference (the device path asks the resident submap for the patch size then: SubmapHip::patchCount)."""
    m = make(ref_period=10.0)                     # the reference is renewed on scan 1 only
    m.T = tx(0.0)
    assert m.add(PTS, PTS, 0.0) and m.add(PTS, PTS, 0.1) and m.flags[1] == 1
    assert m.add(PTS, PTS, 0.2) and m.flags[1] == 0 and m.sm.patch_checks == 1
    pose, calls = m.T.copy(), len(m.icp.calls)
    m.sm.patch_size = 0                           # e.g. carved away, or the active submap changed
    assert not m.add(PTS, PTS, 0.3)
    assert len(m.icp.calls) == calls and np.array_equal(m.T, pose)
