"""Device-resident submap (include/o3s_submap.h) against the CPU oracle.  MI355X only.

The oracle side is the reference's sequence restated step by step — o3d_slam::transform, append, cropper pose,
voxelizeWithinCroppingVolume, crop, open3dToPointmatcher, initReference — on host arrays.  The reference's voxel output
order is its unordered_map's iteration order (unspecified); both sides here keep the voxel part in ascending (z, y, x)
voxel-index order, so the map after every insert must agree BIT FOR BIT, and so must the ICP pose on the cropped patch."""
import math

import numpy as np
import pytest

from oracle import oracle as orc
from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, Submap
from open3d_slam_advanced_rss_2024_public_amd import cloud_ops as co
from open3d_slam_advanced_rss_2024_public_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["hinted", "measured", "hint_miss"], autouse=True)
def index_range_path(request, monkeypatch):
    """Every test runs three times: with the voxel index range hinted by the cropping volume (one read-back per pipeline) on the
    PRODUCT library, and twice on the test-hook build (libo3dslam_icp_hip_hooks.so — the product carries no hook): with the range
    measured on the device and read back (the path for unbounded volumes, forced by O3S_NO_HINT), and with a hint that nothing
    fits in (O3S_HINT_MISS), so that the device's status word trips and the call is repeated on the measuring path.  Same bits."""
    from open3d_slam_advanced_rss_2024_public_amd import _lib

    monkeypatch.delenv("O3S_NO_HINT", raising=False)
    monkeypatch.delenv("O3S_HINT_MISS", raising=False)
    if request.param == "hinted":
        yield request.param
        return
    monkeypatch.setenv("O3S_NO_HINT" if request.param == "measured" else "O3S_HINT_MISS", "1")
    with _lib.variant("hooks"):
        yield request.param


def oracle_insert(map_p, map_n, scan_p, scan_n, T, voxel, kind, params):
    tp, tn = orc.transform_cloud(T, scan_p, scan_n)
    p = tp if map_p is None else np.concatenate([map_p, tp])
    n = None if scan_n is None else (tn if map_n is None else np.concatenate([map_n, tn]))
    if not voxel > 0:
        return p, n
    c = orc.make_cropper(kind, *params, centre=T[:3, 3])
    op, on, oi = orc.voxelize_within_crop(c, voxel, p, n)
    passthrough = oi[:, 0] == np.iinfo(np.int32).min
    k = int(passthrough.sum())
    assert passthrough[:k].all()
    order = np.lexsort((oi[k:, 0], oi[k:, 1], oi[k:, 2])) + k   # canonical order of the voxel part
    idx = np.concatenate([np.arange(k), order])
    return op[idx], (None if on is None else on[idx])


def trajectory(world_seed=3, n_scans=6, n_pts=20000):
    world = syn.make_world(9000.0, seed=world_seed)
    out = []
    for k in range(n_scans):
        pos = np.array([-6.0 + 2.5 * k, 1.0 + 0.7 * k, 1.5])
        T = syn.make_T(syn.rot_axis_angle([0, 0, 1], 0.3 * k), pos)
        sp, sn = syn.make_scan(world, n_pts, T, radius=12.0, sigma=0.01, seed=100 + k)
        out.append((sp.astype(np.float64), sn.astype(np.float64), T))
    return out


@pytest.mark.parametrize("with_normals", [True, False])
def test_insert_sequence_bit_exact(with_normals):
    voxel, kind, params = 0.15, "MaxRadius", (10.0, 0.0, 0.0)
    sm = Submap(voxel, co.croppingVolumeFactory(kind, *params))
    mp = mn = None
    sizes = []
    for sp, sn, T in trajectory():
        sn = sn if with_normals else None
        assert sm.insertScan(sp, sn, T)
        mp, mn = oracle_insert(mp, mn, sp, sn, T, voxel, kind, params)
        gp, gn = sm.getMapPointCloud()
        assert len(sm) == mp.shape[0]
        assert np.array_equal(gp, mp)
        if with_normals:
            assert np.array_equal(gn, mn)
        else:
            assert gn is None
        sizes.append(len(sm))
    assert sizes[-1] > sizes[0] > 1000   # the map grows along the trajectory; older parts pass through unvoxelised


@pytest.mark.parametrize("with_normals", [True, False])
def test_merge_insert_equals_the_sort_based_insert_and_the_oracle(with_normals, monkeypatch, index_range_path):
    """The map is kept in voxel order between inserts and a scan is MERGED into it (voxel_insert_merge_dev; the reference's TODO at
    Submap.cpp:89-92) — out and back along a line, so that points left behind come back inside the volume and the merge has to give
    way to the sort (checked on the device): same map as the sort-only path, bit for bit, and as the oracle."""
    voxel, kind, params = 0.15, "MaxRadius", (9.0, 0.0, 0.0)
    world = syn.make_world(9000.0, seed=5)
    xs = [-8.0, -5.0, -2.0, 1.0, 4.0, 7.0, 10.0, 13.0, 10.0, 6.0, 2.0, -2.0, -6.0]     # out 21 m (> the 18 m the volume spans), then back
    traj = []
    for k, x in enumerate(xs):
        T = syn.make_T(syn.rot_axis_angle([0, 0, 1], 0.2 * k), np.array([x, 0.5, 1.5]))
        sp, sn = syn.make_scan(world, 15000, T, radius=8.0, sigma=0.01, seed=400 + k)
        traj.append((sp.astype(np.float64), sn.astype(np.float64) if with_normals else None, T))
    a = Submap(voxel, co.croppingVolumeFactory(kind, *params))
    mp = mn = None
    for sp, sn, T in traj:
        assert a.insertScan(sp, sn, T)
        mp, mn = oracle_insert(mp, mn, sp, sn, T, voxel, kind, params)
        gp, gn = a.getMapPointCloud()
        assert np.array_equal(gp, mp)
        if with_normals:
            assert np.array_equal(gn, mn)
    merged, sorted_, fell_back = a.insert_stats()
    if index_range_path == "hinted":
        assert merged >= 5 and fell_back >= 1 and merged + sorted_ == len(traj)     # both routes were taken
    elif index_range_path == "measured":
        assert merged == 0 and fell_back == 0            # no bounded index range, no merge
    from open3d_slam_advanced_rss_2024_public_amd import _lib

    monkeypatch.setenv("O3S_INSERT_SORT", "1")
    with _lib.variant("hooks"):   # the sort-only insert is a hook of the test build
        b = Submap(voxel, co.croppingVolumeFactory(kind, *params))
        for sp, sn, T in traj:
            assert b.insertScan(sp, sn, T)
        assert b.insert_stats()[0] == 0 and b.insert_stats()[2] == 0
    pa, na = a.getMapPointCloud()
    pb, nb = b.getMapPointCloud()
    assert np.array_equal(pa, pb) and (na is None) == (nb is None) and (na is None or np.array_equal(na, nb))


def test_merge_insert_of_a_scan_the_reference_enters_twice(index_range_path):
    """An (almost-)identity pose makes the reference's transform() return the cloud AND its transformed copy (helpers.cpp:285-288):
    the scan enters the map twice.  Here into a map that is already in voxel order, i.e. through the merge."""
    voxel, kind, params = 0.15, "MaxRadius", (9.0, 0.0, 0.0)
    world = syn.make_world(9000.0, seed=6)
    T0 = syn.make_T(syn.rot_axis_angle([0, 0, 1], 0.1), np.array([-0.5, 0.2, 0.0]))
    s0 = syn.make_scan(world, 12000, syn.make_T(None, np.array([-0.5, 0.2, 1.5])), radius=7.0, sigma=0.01, seed=51)
    s1 = syn.make_scan(world, 12000, syn.make_T(None, np.array([0.0, 0.0, 1.5])), radius=7.0, sigma=0.01, seed=52)
    a = Submap(voxel, co.croppingVolumeFactory(kind, *params))
    mp = mn = None
    for (sp, sn), T in ((s0, T0), (s1, np.eye(4))):
        sp, sn = sp.astype(np.float64), sn.astype(np.float64)
        assert a.insertScan(sp, sn, T)
        mp, mn = oracle_insert(mp, mn, sp, sn, T, voxel, kind, params)
        gp, gn = a.getMapPointCloud()
        assert np.array_equal(gp, mp) and np.array_equal(gn, mn)
    if index_range_path == "hinted":
        assert a.insert_stats() == (1, 1, 0)      # the first insert sorts, the doubled one is merged


def test_reserve_keeps_the_map_and_later_inserts_give_the_same_bits():
    """o3s_submap_reserve (room for SubmapParameters::maxNumPoints_ up front) moves the arrays of a map that already holds
    points: the contents survive, and the inserts that follow give the same map as without it."""
    voxel, kind, params = 0.15, "MaxRadius", (10.0, 0.0, 0.0)
    a = Submap(voxel, co.croppingVolumeFactory(kind, *params))
    b = Submap(voxel, co.croppingVolumeFactory(kind, *params))
    b.reserve(50_000)                      # on an empty map
    traj = trajectory()
    for k, (sp, sn, T) in enumerate(traj):
        assert a.insertScan(sp, sn, T) and b.insertScan(sp, sn, T)
        if k == 1:
            before = b.getMapPointCloud()
            b.reserve(3_000_000)           # on a map with points: both ping-pong arrays and the work area move
            after = b.getMapPointCloud()
            assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
    pa, na = a.getMapPointCloud()
    pb, nb = b.getMapPointCloud()
    assert np.array_equal(pa, pb) and np.array_equal(na, nb)
    with pytest.raises(RuntimeError):
        b.reserve(-1)


def test_identity_pose_enters_the_scan_twice_like_the_reference():
    """helpers.cpp:285-288: for max|T - I| < 1e-4 the output starts as a copy of the input and the transformed points
    are appended on top — restated on both sides."""
    sp, sn, _ = trajectory(n_scans=1, n_pts=5000)[0]
    T = np.eye(4)
    T[0, 3] = 5e-5
    sm = Submap(0.0, co.croppingVolumeFactory("MaxRadius", 10.0))   # voxel size 0: "Not voxelizing the map"
    sm.insertScan(sp, sn, T)
    mp, mn = oracle_insert(None, None, sp, sn, T, 0.0, "MaxRadius", (10.0, 0.0, 0.0))
    gp, gn = sm.getMapPointCloud()
    assert len(sm) == 2 * sp.shape[0] == mp.shape[0]
    assert np.array_equal(gp, mp) and np.array_equal(gn, mn)
    assert np.array_equal(gp[:5000], sp)


def test_empty_scan_and_mixed_normals():
    sm = Submap(0.1, co.croppingVolumeFactory("MaxRadius", 10.0))
    assert sm.insertScan(np.zeros((0, 3)), None, np.eye(4)) and len(sm) == 0
    sp, sn, T = trajectory(n_scans=1, n_pts=2000)[0]
    sm.insertScan(sp, sn, T)
    with pytest.raises(RuntimeError):
        sm.insertScan(sp, None, T)


@pytest.mark.parametrize("kind,params", [("MaxRadius", (8.0, 0.0, 0.0)), ("Cylinder", (9.0, -1.0, 4.0))])
def test_set_reference_equals_host_path(kind, params):
    """cropSubmap + open3dToPointmatcher + initReference on the device == the same steps through host memory."""
    voxel = 0.12
    sm = Submap(voxel, co.croppingVolumeFactory("MaxRadius", 11.0))
    traj = trajectory(n_scans=5, n_pts=30000)
    for sp, sn, T in traj[:-1]:
        sm.insertScan(sp, sn, T)
    sp, sn, T_gt = traj[-1]
    T_init = syn.perturb_pose(T_gt, 0.08, 1.5, seed=4)
    cfg = IcpConfig()
    a, b = ICP(cfg), ICP(cfg)
    n_patch = sm.set_reference(co.croppingVolumeFactory(kind, *params), T_gt, a)
    # host path: download, oracle crop + conversion, init_reference through host buffers
    mp, mn = sm.getMapPointCloud()
    mask = orc.crop_mask(orc.make_cropper(kind, *params, centre=T_gt[:3, 3]), mp)
    assert n_patch == int(mask.sum()) and 1000 < n_patch < len(sm)
    xyzw, n32 = orc.o3d_to_pm(mp[mask], mn[mask])
    assert b.init_reference(xyzw[:, :3], n32)
    assert np.array_equal(a.reference_mean(), b.reference_mean())
    scan32, scan_n32 = sp.astype(np.float32), sn.astype(np.float32)
    Ta = a.compute(scan32, scan_n32, T_init)
    Tb = b.compute(scan32, scan_n32, T_init)
    assert np.array_equal(Ta, Tb) and a.stats.iterations == b.stats.iterations
    o = orc.OracleIcp(orc.OracleConfig(), threads=4)
    assert o.init_reference(xyzw[:, :3], n32) == orc.OK
    To, code = o.compute(scan32, scan_n32, T_init, raise_on_error=False)
    assert code == orc.OK
    dt, ang = orc.pose_error(To, Ta)
    assert np.linalg.norm(dt) <= 1e-5 and ang <= 1e-5
    dgt, agt = orc.pose_error(T_gt, Ta)
    assert np.linalg.norm(dgt) < 0.02 and agt < 0.01   # and it registers the scan


def test_empty_patch_is_reported():
    sm = Submap(0.1, co.croppingVolumeFactory("MaxRadius", 10.0))
    sp, sn, T = trajectory(n_scans=1, n_pts=2000)[0]
    sm.insertScan(sp, sn, T)
    far = np.eye(4)
    far[:3, 3] = 1e4
    with pytest.raises(RuntimeError, match="empty"):
        sm.set_reference(co.croppingVolumeFactory("MaxRadius", 5.0), far, ICP(IcpConfig()))


def test_upload_roundtrip():
    rng = np.random.default_rng(0)
    p, n = rng.normal(size=(777, 3)), rng.normal(size=(777, 3))
    sm = Submap(0.1, co.croppingVolumeFactory("MaxRadius", 10.0))
    sm.setMapPointCloud(p, n)
    gp, gn = sm.getMapPointCloud()
    assert np.array_equal(gp, p) and np.array_equal(gn, n)


# ---------------------------------------------------------------------------------------------------------------
# device-resident pre-processed scan (include/o3s_scan.h) and the whole per-scan loop in HBM
# ---------------------------------------------------------------------------------------------------------------
def oracle_preprocess(sp, sn, wide, voxel, narrow):
    m = orc.crop_mask(orc.make_cropper(*wide), sp)
    p, n = sp[m], sn[m]
    if voxel > 0:
        p, n, idx = orc.voxel_downsample_o3d(voxel, p, n)
        order = np.lexsort((idx[:, 0], idx[:, 1], idx[:, 2]))   # canonical voxel order (Open3D's is unspecified)
        p, n = p[order], n[order]
    m2 = orc.crop_mask(orc.make_cropper(*narrow), p)
    return (p, n), (p[m2], n[m2])


@pytest.mark.parametrize("wide", [("MaxRadius", 11.0), ("Cylinder", 10.0, -1.5, 4.0), ("MinMaxRadius", 2.0, 11.0), ("MinRadius", 3.0)])
@pytest.mark.parametrize("voxel", [0.12, 0.0])
def test_preprocess_bit_exact(voxel, wide):
    from open3d_slam_advanced_rss_2024_public_amd import ProcessedScan

    sp, sn, _ = trajectory(n_scans=1, n_pts=40000)[0]
    narrow = ("Cylinder", 8.0, -1.2, 3.0)
    ps = ProcessedScan()
    n_merge, n_match = ps.preprocess(co.croppingVolumeFactory(*wide), voxel, co.croppingVolumeFactory(*narrow), sp, sn)
    (mp, mn), (np_, nn) = oracle_preprocess(sp, sn, wide, voxel, narrow)
    gp, gn = ps.merge
    assert n_merge == mp.shape[0] and np.array_equal(gp, mp) and np.array_equal(gn, mn)
    hp, hn = ps.match
    assert n_match == np_.shape[0] and np.array_equal(hp, np_) and np.array_equal(hn, nn)
    assert 0 < n_match < n_merge <= sp.shape[0]
    with pytest.raises(RuntimeError, match="normals"):
        ps.preprocess(co.croppingVolumeFactory(*wide), voxel, co.croppingVolumeFactory(*narrow), sp, None)


def test_whole_scan_loop_stays_on_device_and_matches_host_path():
    """raw scan -> preprocess -> reading -> ICP against the submap patch -> insert the merge cloud: every step on the
    device equals the same step done through host buffers (oracle pre-processing, host-pointer ICP, host insert)."""
    from open3d_slam_advanced_rss_2024_public_amd import ProcessedScan

    voxel_map, voxel_scan = 0.12, 0.1
    wide, narrow = ("MaxRadius", 11.0), ("MaxRadius", 9.0)
    dev_map = Submap(voxel_map, co.croppingVolumeFactory("MaxRadius", 11.0))
    host_map = Submap(voxel_map, co.croppingVolumeFactory("MaxRadius", 11.0))
    icp_dev, icp_host = ICP(IcpConfig()), ICP(IcpConfig())
    ps = ProcessedScan()
    T_prev = None
    for k, (sp, sn, T_gt) in enumerate(trajectory(n_scans=5, n_pts=30000)):
        ps.preprocess(co.croppingVolumeFactory(*wide), voxel_scan, co.croppingVolumeFactory(*narrow), sp, sn)
        (mp, mn), (qp, qn) = oracle_preprocess(sp, sn, wide, voxel_scan, narrow)
        if k == 0:
            T_dev = T_host = T_gt     # the first scan seeds the map at its given pose
        else:
            T_init = syn.perturb_pose(T_gt, 0.06, 1.0, seed=10 + k)
            cropper = co.croppingVolumeFactory("MaxRadius", 10.0)
            dev_map.set_reference(cropper, T_prev, icp_dev)
            ps.set_reading(icp_dev)
            T_dev = icp_dev.compute_resident(T_init)
            hp, hn = host_map.getMapPointCloud()
            mask = orc.crop_mask(orc.make_cropper("MaxRadius", 10.0, centre=np.asarray(T_prev)[:3, 3]), hp)
            xyzw, n32 = orc.o3d_to_pm(hp[mask], hn[mask])
            assert icp_host.init_reference(xyzw[:, :3], n32)
            q32, qn32 = orc.o3d_to_pm(qp, qn)
            T_host = icp_host.compute(q32[:, :3], qn32, T_init)
            assert np.array_equal(T_dev, T_host) and icp_dev.stats.iterations == icp_host.stats.iterations
            dgt, agt = orc.pose_error(T_gt, T_dev)
            assert np.linalg.norm(dgt) < 0.03 and agt < 0.01
        dev_map.insertProcessed(ps, np.asarray(T_dev, np.float64))
        host_map.insertScan(mp, mn, np.asarray(T_host, np.float64))
        a, an = dev_map.getMapPointCloud()
        b, bn = host_map.getMapPointCloud()
        assert np.array_equal(a, b) and np.array_equal(an, bn)
        T_prev = np.asarray(T_dev, np.float64)
    assert len(dev_map) > 20000


def test_insert_whose_completion_is_pending_gives_the_map_of_the_insert_that_waits(monkeypatch, hooks_lib, index_range_path):
    """o3s_submap_insert_processed returns with the merge insert enqueued and nobody waiting for its counts; whatever call comes
    next on the submap completes it.  Same maps, point for point, as with the insert that waits (O3S_INSERT_EAGER=1, hooks build) —
    on a drive that goes out and comes back, so that pass-through points re-enter the volume and a PENDING merge has to give way to
    the sort-based path while it is completed — whichever call does the completing: size, download, the next insert, set_reference,
    a clone.  The bounds that are answered without waiting hold the size that comes out."""
    from open3d_slam_advanced_rss_2024_public_amd import ProcessedScan

    wide, narrow = ("MaxRadius", 9.0), ("MaxRadius", 8.0)
    world = syn.make_world(60000.0, seed=11)
    poses = [syn.corridor_pose(world, k, 1.5) for k in (0, 1, 2, 3, 4, 5, 6, 5, 4, 3, 2, 1, 0, 1, 2)]     # out, back over old ground, out again
    sweeps = [syn.make_lidar_scan(world, T, 32, 512, max_range=40.0, sigma=0.01, seed=500 + k) for k, T in enumerate(poses)]
    maps, stats, bounds_seen = {}, {}, []
    for mode in ("eager", "pending"):
        if mode == "eager":
            monkeypatch.setenv("O3S_INSERT_EAGER", "1")
        else:
            monkeypatch.delenv("O3S_INSERT_EAGER", raising=False)
        m = Submap(0.1, co.croppingVolumeFactory(*wide))
        icp = ICP(IcpConfig())
        ps = [ProcessedScan(), ProcessedScan()]
        out = []
        for k, ((sp, sn), T) in enumerate(zip(sweeps, poses)):
            sc = ps[k & 1]
            sc.preprocess(co.croppingVolumeFactory(*wide), 0.1, co.croppingVolumeFactory(*narrow), sp.astype(np.float64), sn.astype(np.float64))
            m.insertProcessed(sc, np.asarray(T, np.float64))
            lo, hi = m.size_bounds()                       # never waits
            what = k % 5                                   # who completes the insert
            if what == 0:
                n = len(m)
            elif what == 1:
                n = len(m.getMapPointCloud()[0])
            elif what == 2:
                n = None                                   # the next insert does
            elif what == 3:
                m.set_reference(co.croppingVolumeFactory("MaxRadius", 8.0), np.asarray(T, np.float64), icp)
                n = len(m)
            else:
                c = m.clone()
                n = len(c)
                assert np.array_equal(c.getMapPointCloud()[0], m.getMapPointCloud()[0])
            if n is not None:
                assert lo <= n <= hi and lo >= 1
                assert m.size_bounds() == (n, n)           # nothing pending any more: exact
                if mode == "pending":
                    bounds_seen.append(hi > lo)
            out.append(m.getMapPointCloud() if n is not None else None)
        maps[mode] = (out, m.getMapPointCloud())
        stats[mode] = m.insert_stats()
    if index_range_path == "hinted":     # (the measuring path has no merge insert, hence nothing to leave pending)
        assert any(bounds_seen), "no insert was ever pending: the test did not test it"
    for a, b in zip(maps["eager"][0], maps["pending"][0]):
        assert (a is None) == (b is None)
        if a is not None:
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(maps["eager"][1][0], maps["pending"][1][0]) and np.array_equal(maps["eager"][1][1], maps["pending"][1][1])
    assert stats["eager"] == stats["pending"], stats
    if index_range_path == "hinted":
        assert stats["pending"][0] >= 1 and stats["pending"][2] >= 1, stats      # (merged, sorted, fell back): the return trip made a merge give way


def test_two_submaps_with_pending_inserts_on_one_thread(monkeypatch, hooks_lib, index_range_path):
    """The counts of a pending insert travel to ONE mailbox slot per host thread.  A thread that inserts into a second submap while the
    first insert is still pending (the overlap scans a new submap is seeded with) must neither mix the two posts up nor lose one:
    alternating inserts into two submaps, nothing looked at until the end, give the maps of the inserts that wait."""
    from open3d_slam_advanced_rss_2024_public_amd import ProcessedScan

    wide, narrow = ("MaxRadius", 9.0), ("MaxRadius", 8.0)
    world = syn.make_world(60000.0, seed=12)
    poses = [syn.corridor_pose(world, k, 1.0) for k in range(8)]
    sweeps = [syn.make_lidar_scan(world, T, 32, 512, max_range=40.0, sigma=0.01, seed=700 + k) for k, T in enumerate(poses)]
    maps = {}
    for mode in ("eager", "pending"):
        if mode == "eager":
            monkeypatch.setenv("O3S_INSERT_EAGER", "1")
        else:
            monkeypatch.delenv("O3S_INSERT_EAGER", raising=False)
        a, b = Submap(0.1, co.croppingVolumeFactory(*wide)), Submap(0.12, co.croppingVolumeFactory(*wide))
        scans = [ProcessedScan() for _ in range(3)]
        pend_seen = 0
        for k, ((sp, sn), T) in enumerate(zip(sweeps, poses)):
            sc = scans[k % 3]
            sc.preprocess(co.croppingVolumeFactory(*wide), 0.1, co.croppingVolumeFactory(*narrow), sp.astype(np.float64), sn.astype(np.float64))
            a.insertProcessed(sc, np.asarray(T, np.float64))          # pending on a ...
            b.insertProcessed(sc, np.asarray(T, np.float64))          # ... and now on b as well: the slot has to be free first
            la, lb = a.size_bounds(), b.size_bounds()
            pend_seen += int(la[1] > la[0]) + int(lb[1] > lb[0])
        maps[mode] = (a.getMapPointCloud(), b.getMapPointCloud(), a.insert_stats(), b.insert_stats())
        if mode == "pending" and index_range_path == "hinted":
            assert pend_seen >= 4
    for x, y in zip(maps["eager"][:2], maps["pending"][:2]):
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
    assert maps["eager"][2:] == maps["pending"][2:]


# ---------------------------------------------------------------------------------------------------------------
# space carving
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("with_normals", [True, False])
def test_carve_matches_oracle(with_normals):
    """A 'ghost' object standing in free space between the sensor and the walls must be carved; the decision per map
    point, the survivors and their order equal the oracle's restatement of getIdxsOfCarvedPoints + removeByIds."""
    voxel = 0.12
    traj = trajectory(n_scans=3, n_pts=30000)
    sm = Submap(voxel, co.croppingVolumeFactory("MaxRadius", 11.0))
    for sp, sn, T in traj[:2]:
        sm.insertScan(sp, sn if with_normals else None, T)
    mp, mn = sm.getMapPointCloud()
    # add ghost points (a small box of points hanging in free space) straight into the resident map
    sp, sn, T = traj[2]
    rng = np.random.default_rng(1)
    c = T[:3, 3] + T[:3, :3] @ np.array([1.0, 1.0, -0.5])      # below the horizon: rays to the floor cross it
    ghost = c + rng.uniform(-0.15, 0.15, (200, 3))
    gn = np.tile(T[:3, :3] @ (np.array([-1.0, -1.0, 0.5]) / 1.5), (200, 1))    # facing the sensor
    mp2 = np.concatenate([mp, ghost])
    mn2 = np.concatenate([mn, gn]) if with_normals else None
    sm.setMapPointCloud(mp2, mn2)
    raw = sp[:20000]
    n_removed = sm.carve(raw, T, voxel_size=0.1, max_raytracing_length=20.0, truncation_distance=0.1, min_dot_product_with_normal=0.5)
    # oracle: cropper still sits at the pose of the previous insert (Submap.cpp:66-86)
    scan_map, _ = orc.transform_cloud(T, raw, None)
    subset = orc.crop_mask(orc.make_cropper("MaxRadius", 11.0, centre=traj[1][2][:3, 3]), mp2)
    rm = orc.carve(scan_map, mp2, mn2, T[:3, 3], 0.1, 20.0, 0.1, 0.5, subset=subset)
    assert n_removed == int(rm.sum()) and n_removed >= 100          # most of the ghost goes
    gp, gnn = sm.getMapPointCloud()
    assert np.array_equal(gp, mp2[~rm])
    if with_normals:
        assert np.array_equal(gnn, mn2[~rm])
    assert rm[len(mp):].sum() >= 100


def test_carve_empty_inputs():
    sm = Submap(0.1, co.croppingVolumeFactory("MaxRadius", 10.0))
    assert sm.carve(np.zeros((10, 3)) + 1.0, np.eye(4)) == 0      # empty map
    sp, sn, T = trajectory(n_scans=1, n_pts=2000)[0]
    sm.insertScan(sp, sn, T)
    n0 = len(sm)
    assert sm.carve(np.zeros((0, 3)), T) == 0 and len(sm) == n0
    assert sm.carve(np.zeros((1, 3)), T) == 0                     # a return at the sensor origin removes nothing


def test_coloured_scans_round_trip_through_the_resident_map():
    """Colours in the resident submap (o3s_submap_insert_scan_colored): after every insert the map's colours equal the
    oracle's restatement of transform (colours copied) + `+=` + voxelizeWithinCroppingVolume (last colour per voxel); a scan
    without colours clears them, as Open3D's operator+= does."""
    voxel, kind, params = 0.15, "MaxRadius", (10.0, 0.0, 0.0)
    sm = Submap(voxel, co.croppingVolumeFactory(kind, *params))
    rng = np.random.default_rng(8)
    mp = mn = mc = None
    for k, (sp, sn, T) in enumerate(trajectory(n_scans=4)):
        sc = rng.uniform(0, 1, sp.shape)
        assert sm.insertScanColored(sp, sn, sc, T)
        tp, tn = orc.transform_cloud(T, sp, sn)
        p = tp if mp is None else np.concatenate([mp, tp])
        n = tn if mn is None else np.concatenate([mn, tn])
        c = sc if mc is None else np.concatenate([mc, sc])
        cr = orc.make_cropper(kind, *params, centre=T[:3, 3])
        op, on, oi = orc.voxelize_within_crop(cr, voxel, p, n)
        oc, _ = orc.voxelize_attrs(0, cr, voxel, p, c, None)
        kk = int((oi[:, 0] == np.iinfo(np.int32).min).sum())
        order = np.concatenate([np.arange(kk), np.lexsort((oi[kk:, 0], oi[kk:, 1], oi[kk:, 2])) + kk])
        mp, mn, mc = op[order], on[order], oc[order]
        gp, gn = sm.getMapPointCloud()
        assert sm.hasColors() and np.array_equal(gp, mp) and np.array_equal(gn, mn) and np.array_equal(sm.getMapColors(), mc)
    sp, sn, T = trajectory(n_scans=5)[4]
    sm.insertScan(sp, sn, T)                      # a scan without colours: the map's colours are gone (PointCloud::operator+=)
    assert not sm.hasColors()
    with pytest.raises(RuntimeError):
        sm.getMapColors()


def test_trim_gives_back_everything_but_the_map_and_the_submap_stays_usable():
    """A submap that stops being the active one (SubmapCollection.cpp:94-162) keeps its map cloud and nothing else
    (o3s_submap_trim: the spare ping-pong arrays and the work area of a reserved submap are ~10x the map of a submap closed early);
    the map bits do not move, and a later insert — a buffered scan, a re-activation — still equals the oracle's."""
    world = syn.make_world(3000.0, seed=5)
    voxel, kind, params = 0.1, "MaxRadius", (12.0,)
    a = Submap(voxel, co.croppingVolumeFactory(kind, *params))
    a.reserve(400_000 + 262_144)                     # what SubmapCollectionHip::createNewSubmap reserves at the default limits
    mp = mn = None
    traj = []
    for k in range(4):
        T = syn.make_T(syn.rot_axis_angle([0, 0, 1], 0.1 * k), np.array([0.4 * k, 0.2, 1.5]))
        sp, sn = syn.make_scan(world, 12000, T, radius=8.0, sigma=0.01, seed=700 + k)
        traj.append((sp.astype(np.float64), sn.astype(np.float64), T))
    for sp, sn, T in traj[:3]:
        assert a.insertScan(sp, sn, T)
        mp, mn = oracle_insert(mp, mn, sp, sn, T, voxel, kind, params)
    before = a.device_bytes()
    p0, n0 = a.getMapPointCloud()
    a.trim()
    after = a.device_bytes()
    assert after < before / 4 and after >= len(a) * 48
    p1, n1 = a.getMapPointCloud()
    assert np.array_equal(p0, p1) and np.array_equal(n0, n1) and np.array_equal(p1, mp)
    sp, sn, T = traj[3]
    assert a.insertScan(sp, sn, T)                   # buffers come back on demand
    mp, mn = oracle_insert(mp, mn, sp, sn, T, voxel, kind, params)
    p2, n2 = a.getMapPointCloud()
    assert np.array_equal(p2, mp) and np.array_equal(n2, mn)


def test_hand_over_moves_everything_but_the_map_to_the_next_submap(index_range_path):
    """SubmapCollection's switch to a NEW submap (SubmapCollection.cpp:150-162, 216-239): the closed submap keeps its map in arrays of
    its own size and the successor takes every other buffer over as it is (o3s_submap_hand_over) — no byte is freed or allocated
    beyond the closed map's tight copy, the closed map's bits do not move, and both submaps go on equal to the oracle's: the
    successor through the buffered scans and further inserts, the closed one through a late insert (its buffers come back on demand)."""
    world = syn.make_world(3000.0, seed=6)
    voxel, kind, params = 0.1, "MaxRadius", (12.0,)
    a = Submap(voxel, co.croppingVolumeFactory(kind, *params))
    a.reserve(400_000 + 262_144)
    traj = []
    for k in range(7):
        T = syn.make_T(syn.rot_axis_angle([0, 0, 1], 0.1 * k), np.array([0.5 * k, 0.2, 1.5]))
        sp, sn = syn.make_scan(world, 12000, T, radius=8.0, sigma=0.01, seed=900 + k)
        traj.append((sp.astype(np.float64), sn.astype(np.float64), T))
    mp = mn = None
    for sp, sn, T in traj[:3]:
        assert a.insertScan(sp, sn, T)
        mp, mn = oracle_insert(mp, mn, sp, sn, T, voxel, kind, params)
    p0, n0 = a.getMapPointCloud()
    held = a.device_bytes()
    b = Submap(voxel, co.croppingVolumeFactory(kind, *params))
    assert b.device_bytes() == 0
    a.hand_over(b)
    assert a.device_bytes() <= 2 * (len(a) * 24 + 4096) + 256                 # the map (points + normals) and the 128-byte pose staging
    assert a.device_bytes() + b.device_bytes() <= held + 2 * (len(a) * 24 + 4096)   # nothing new but the tight copy
    assert b.device_bytes() >= held - 2 * (len(a) * 24 + 4096) - 256 and len(b) == 0
    p1, n1 = a.getMapPointCloud()
    assert np.array_equal(p0, p1) and np.array_equal(n0, n1) and np.array_equal(p1, mp) and np.array_equal(n1, mn)
    # the successor: the buffered scans (the last ones of the closed submap) and new ones, equal to the oracle's on its own history
    bp = bn = None
    for sp, sn, T in traj[1:6]:
        assert b.insertScan(sp, sn, T)
        bp, bn = oracle_insert(bp, bn, sp, sn, T, voxel, kind, params)
    q, qn = b.getMapPointCloud()
    assert np.array_equal(q, bp) and np.array_equal(qn, bn)
    if index_range_path == "hinted":
        assert b.insert_stats()[0] >= 3                                       # and it merges like any reserved submap
    # the closed one is still a submap: a late insert equals the oracle's too
    sp, sn, T = traj[6]
    assert a.insertScan(sp, sn, T)
    mp, mn = oracle_insert(mp, mn, sp, sn, T, voxel, kind, params)
    p2, n2 = a.getMapPointCloud()
    assert np.array_equal(p2, mp) and np.array_equal(n2, mn)
    # a submap that already holds points cannot take buffers over this way
    with pytest.raises(Exception):
        a.hand_over(b)
