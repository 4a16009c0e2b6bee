"""Parity of the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.  MI355X only.

Bars (BASELINE.json north_star): correspondence ids / squared distances and the trim limit bit-exact; final pose within
1e-4 m / 1e-4 rad of the CPU path (we assert far tighter where the arithmetic allows it)."""
import math
import os

import numpy as np
import pytest

from oracle import oracle as orc
from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, ConvergenceError, TransformationError
from open3d_slam_advanced_rss_2024_public_amd import synthetic as syn

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4     # north_star: pose within 1e-4 m / 1e-4 rad of the reference CPU ICP
POSE_TOL_RAD = 1e-4


def both(cfg_kwargs, ocfg_kwargs):
    return ICP(IcpConfig(**cfg_kwargs)), orc.OracleIcp(orc.OracleConfig(**ocfg_kwargs), threads=8)


def yaml_pair(**over):
    g = dict(max_dist=0.5, trim_ratio=0.9, max_normal_angle=1.57, use_differential=True, min_diff_rot=0.001,
             min_diff_trans=0.01, smooth_length=3, max_iters=15, counter_first=False)
    g.update(over)
    o = dict(g)
    for k in ("trim_ratio", "max_normal_angle", "max_dist_outlier"):
        if k in o and o[k] is None:
            o[k] = -1.0
    extra = {k: g.pop(k) for k in list(g) if k in ("grid_cell", "sort_queries", "use_graph", "match_stats", "epsilon")}
    for k in extra:
        o.pop(k)
    gi = dict(g)
    gi.update(extra)
    return both(gi, o)


def assert_pose_close(Tg, To, tol_m=POSE_TOL_M, tol_rad=POSE_TOL_RAD):
    dt, ang = orc.pose_error(To, Tg)
    assert np.linalg.norm(dt) <= tol_m and ang <= tol_rad, (dt, ang)


# ---------------------------------------------------------------------------------------------------------------
# matcher
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("max_dist,grid_cell,sort", [(0.2, 0.0, True), (0.2, 0.05, False), (float("inf"), 0.0, True), (1.0, 0.13, True)])
def test_find_closests_bit_exact(max_dist, grid_cell, sort):
    rng = np.random.default_rng(11)
    ref = rng.uniform(-2, 2, (20000, 3)).astype(np.float32)
    ref[100] = ref[50]              # exact duplicate -> lowest index wins
    ref[7000:7010] = ref[6000]      # a pile of duplicates in one cell
    nrm = np.zeros_like(ref)
    g, o = both(dict(max_dist=max_dist, grid_cell=grid_cell, sort_queries=sort), dict(max_dist=max_dist))
    assert g.init_reference(ref, nrm) and o.init_reference(ref, nrm) == orc.OK
    assert np.array_equal(g.reference_mean(), o.reference_mean())
    q = rng.uniform(-2.6, 2.6, (30000, 3)).astype(np.float32)
    q[0] = ref[100] - o.reference_mean()
    q[1] = ref[7005] - o.reference_mean()
    q[2] = (50.0, -80.0, 3.0)       # far outside the grid
    ids_g, d_g = g.find_closests(q)
    ids_o, d_o = o.find_closests(q, brute=False)
    assert np.array_equal(ids_g, ids_o)
    assert np.array_equal(d_g.view(np.uint32), d_o.view(np.uint32))
    assert ids_g[0] == 50 and ids_g[1] == 6000
    if math.isfinite(max_dist):
        assert ids_g[2] == -1 and np.isinf(d_g[2])
    else:
        assert np.all(ids_g >= 0)


def test_matcher_init_keeps_the_cloud_as_given_through_the_c_abi():
    """o3s_matcher_init = Matcher::init (LPM/PointMatcher.h:559-561, MatchersImpl.cpp:108-132): the adapter's frame, not the fused
    path's.  The cloud ICP::initReference hands its matcher is already centred with an fp32 mean (ICP.cpp:313-324): its residual
    mean is ~1e-9, many coordinates are small, and subtracting that residual again moves them by an ulp.  ids and d2 must equal
    the oracle's Matcher::init + findClosests on that very cloud, bit for bit — and must DIFFER from what a re-centring index
    returns for the same query, or the test would not see the bug it is here for."""
    rng = np.random.default_rng(21)
    M = 60000
    raw = (rng.normal(size=(M, 3)) * np.array([0.02, 3.0, 0.5]) + np.array([11.3, -4.7, 2.2])).astype(np.float32)
    mean = (raw.astype(np.float64).sum(axis=0) / M).astype(np.float32)
    centred = raw - mean                                    # what referenceFiltered holds when matcher->init sees it
    resid = (centred.astype(np.float64).sum(axis=0) / M).astype(np.float32)
    assert np.any(resid != 0) and np.mean(np.abs(centred[:, 0]) < 0.03) > 0.5
    nrm = np.zeros_like(centred)
    g, o = both(dict(max_dist=0.5), dict(max_dist=0.5))
    assert g.matcher_init(centred, nrm) and o.matcher_init(centred, nrm) == orc.OK
    assert np.array_equal(g.reference_mean(), np.zeros(3, np.float32))
    q = np.concatenate([centred[::7] + rng.normal(scale=0.01, size=(len(centred[::7]), 3)).astype(np.float32),
                        centred[:2000],                                               # self-queries: distance exactly 0
                        rng.uniform(-9, 9, (3000, 3)).astype(np.float32)])           # some beyond maxDist
    ids_g, d_g = g.find_closests(q)
    ids_o, d_o = o.find_closests(q)
    assert np.array_equal(ids_g, ids_o) and np.array_equal(d_g.view(np.uint32), d_o.view(np.uint32))
    n_self = len(centred[::7])
    assert np.array_equal(ids_g[n_self:n_self + 2000], np.arange(2000)) and np.all(d_g[n_self:n_self + 2000] == 0)
    assert (ids_g < 0).any() and (ids_g >= 0).mean() > 0.7
    # the re-centring entry point on the same cloud is NOT the same matcher: the self-queries no longer come back at 0 everywhere
    g2 = ICP(IcpConfig(max_dist=0.5))
    assert g2.init_reference(centred, nrm)
    assert np.any(g2.reference_mean() != 0)
    _, d_re = g2.find_closests(centred[:2000])
    assert np.any(d_re != 0)


@pytest.mark.timeout(120)
@pytest.mark.parametrize("max_dist", [0.5, float("inf")])
def test_non_finite_reading_points_have_no_match_and_hang_nothing(max_dist):
    """A NaN or an infinite coordinate in the reading: the point has no neighbour (every distance test fails), on the GPU as in the
    oracle — also with an unbounded maxDist, where a ring search sized from such a query would walk the whole grid — and the
    registration is the one of the clean points, bit for bit (they are summed in the same order)."""
    pair = syn.make_scan_pair(3000, 30000, 0.1, seed=2)
    xyz, nn = pair.scan_xyz.copy(), pair.scan_normals.copy()
    bad = [5, 100, 777, 2999]
    xyz[5, 0] = np.nan
    xyz[100, 1] = np.inf
    xyz[777, 2] = -np.inf
    xyz[2999] = np.nan
    g, o = yaml_pair(max_dist=max_dist)
    assert g.init_reference(pair.map_xyz, pair.map_normals) and o.init_reference(pair.map_xyz, pair.map_normals) == orc.OK
    Tg = g.compute(xyz, nn, pair.T_init)
    To = o.compute(xyz, nn, pair.T_init)
    assert g.stats.iterations == o.stats.iterations and g.stats.kept_pairs == o.stats.kept_pairs
    assert g.stats.matched_pairs == o.stats.matched_pairs <= 3000 - len(bad)
    assert np.array_equal(g.stats.trace_limit, o.trace_limit) and np.array_equal(g.stats.trace_kept, o.trace_kept)
    assert_pose_close(Tg, To, 1e-6, 1e-6)
    keep = np.ones(3000, bool)
    keep[bad] = False
    g2 = ICP(g.config)
    g2.init_reference(pair.map_xyz, pair.map_normals)
    Tc = g2.compute(xyz[keep], nn[keep], pair.T_init)
    assert np.array_equal(g.stats.trace_kept, g2.stats.trace_kept) and np.array_equal(g.stats.trace_limit, g2.stats.trace_limit)
    assert_pose_close(Tg, Tc, 1e-6, 1e-6)
    # module level: ids -1, squared distance +inf
    q = (xyz @ pair.T_init[:3, :3].T.astype(np.float32) + pair.T_init[:3, 3].astype(np.float32)) - g.reference_mean()
    ids, d2 = g.find_closests(q)
    oi, od = o.find_closests(q)
    assert np.array_equal(ids, oi) and np.array_equal(d2, od)
    assert np.all(ids[bad] == -1) and np.all(np.isinf(d2[bad]))


def test_find_closests_surface_map():
    """The benchmark geometry (thin surfaces, ~1 point per 0.1 m voxel), maxDist 0.5 as in icp.yaml."""
    pair = syn.make_scan_pair(20000, 200000, 0.1, seed=5)
    g, o = yaml_pair()
    g.init_reference(pair.map_xyz, pair.map_normals)
    o.init_reference(pair.map_xyz, pair.map_normals)
    T0 = pair.T_init.copy()
    T0[:3, 3] -= o.reference_mean()
    q, _ = orc.rigid_transform(T0, pair.scan_xyz)
    ids_g, d_g = g.find_closests(q)
    ids_o, d_o = o.find_closests(q)
    assert np.array_equal(ids_g, ids_o) and np.array_equal(d_g.view(np.uint32), d_o.view(np.uint32))
    assert (ids_g >= 0).mean() > 0.9


# ---------------------------------------------------------------------------------------------------------------
# outlier chain + trim limit
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ratio", [0.9, 0.5, 0.1, 1.0, 0.999])
def test_outlier_weights_and_exact_trim_element(ratio):
    rng = np.random.default_rng(12)
    M, N = 5000, 40000
    ref = rng.uniform(-1, 1, (M, 3)).astype(np.float32)
    nref = rng.normal(size=(M, 3)).astype(np.float32)
    nref /= np.linalg.norm(nref, axis=1, keepdims=True)
    g, o = yaml_pair(trim_ratio=ratio, max_dist=0.3)
    g.init_reference(ref, nref)
    o.init_reference(ref, nref)
    q = rng.uniform(-1.2, 1.2, (N, 3)).astype(np.float32)
    nq = rng.normal(size=(N, 3)).astype(np.float32)
    nq /= np.linalg.norm(nq, axis=1, keepdims=True)
    ids, d2 = o.find_closests(q)
    d2[10:5000:7] = d2[3]           # heavy ties around one value
    wg = g.outlier_weights(nq, ids, d2)
    wo = o.outlier_weights(nq, ids, d2)
    assert np.array_equal(wg, wo)
    # no normals on the reading -> the normal filter passes everything (OutlierFiltersImpl.cpp:268-277)
    assert np.array_equal(g.outlier_weights(None, ids, d2), o.outlier_weights(None, ids, d2))


def test_outlier_weights_edge_cases():
    ref = np.random.default_rng(1).uniform(-1, 1, (100, 3)).astype(np.float32)
    n = np.tile(np.array([0, 0, 1], np.float32), (100, 1))
    g, o = yaml_pair()
    g.init_reference(ref, n)
    o.init_reference(ref, n)
    ids = np.full(8, -1, np.int32)
    d2 = np.full(8, np.inf, np.float32)
    with pytest.raises(ConvergenceError):           # Matches.cpp:76-77
        g.outlier_weights(None, ids, d2)
    # pattern of utest/ui/Outliers.cpp:126-152 adapted to Trimmed
    d = np.array([4, 5, 5, 5, 5], np.float32)
    idv = np.arange(5, dtype=np.int32)
    g9, _ = yaml_pair(trim_ratio=0.9, max_normal_angle=None)
    g9.init_reference(ref, n)
    assert np.array_equal(g9.outlier_weights(None, idv, d), [1, 1, 1, 1, 1])
    g1, _ = yaml_pair(trim_ratio=0.1, max_normal_angle=None)
    g1.init_reference(ref, n)
    assert np.array_equal(g1.outlier_weights(None, idv, d), [1, 0, 0, 0, 0])
    g0, o0 = yaml_pair(trim_ratio=None, max_normal_angle=None)
    g0.init_reference(ref, n)
    dd = np.array([1, np.inf, 2], np.float32)
    assert np.array_equal(g0.outlier_weights(None, np.array([0, -1, 3], np.int32), dd), [1, 0, 1])


# ---------------------------------------------------------------------------------------------------------------
# minimiser
# ---------------------------------------------------------------------------------------------------------------
def test_minimize_matches_oracle():
    pair = syn.make_scan_pair(8000, 60000, 0.1, seed=6)
    g, o = yaml_pair()
    g.init_reference(pair.map_xyz, pair.map_normals)
    o.init_reference(pair.map_xyz, pair.map_normals)
    T0 = pair.T_init.copy()
    T0[:3, 3] -= o.reference_mean()
    q, qn = orc.rigid_transform(T0, pair.scan_xyz, pair.scan_normals)
    ids, d2 = o.find_closests(q)
    w = o.outlier_weights(qn, ids, d2)
    Tg, Ag, bg, xg = g.minimize(q, ids, d2, w)
    To, Ao, bo, xo = o.p2plane_step(q, ids, d2, w)
    # per-pair arithmetic is identical fp32; the K-long sums are fp64 on both sides and rounded once
    assert np.allclose(Ag, Ao, rtol=2e-7, atol=0) and np.allclose(bg, bo, rtol=2e-6, atol=1e-9)
    assert np.allclose(xg, xo, rtol=1e-4, atol=1e-8)
    assert_pose_close(Tg, To, 1e-6, 1e-6)
    with pytest.raises(ConvergenceError):            # ErrorMinimizer.cpp:75-77
        g.minimize(q, ids, d2, np.zeros_like(w))


def test_minimize_rank_deficient_branch():
    """icpSingular geometry: A has rank 3; the min-norm branch (PointToPlane.cpp:196-233) must give t_z = 1 exactly."""
    nX = 10
    d = np.float32(0.1)
    pts = np.array([[d * x - 0.5, d * y - 0.5, 0] for x in range(nX) for y in range(nX)], np.float32)
    ref = pts.copy()
    ref[:, 2] = 1.0
    n = np.tile(np.array([0, 0, 1], np.float32), (100, 1))
    g, o = yaml_pair(max_dist=float("inf"), trim_ratio=1.0, max_normal_angle=None)
    g.init_reference(ref, n)
    o.init_reference(ref, n)
    q = pts - o.reference_mean()
    ids, d2 = g.find_closests(q)
    ido, d2o = o.find_closests(q)
    assert np.array_equal(ids, ido) and np.array_equal(d2, d2o)
    w = np.ones(100, np.float32)
    Tg, Ag, bg, xg = g.minimize(q, ids, d2, w)
    To, Ao, bo, xo = o.p2plane_step(q, ids, d2, w)
    assert np.array_equal(Ag, Ao) and np.array_equal(bg, bo)
    assert np.allclose(xg, xo, atol=1e-6) and abs(Tg[2, 3] - 1.0) < 1e-6 and np.allclose(Tg[:3, :3], np.eye(3), atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------
# fused path
# ---------------------------------------------------------------------------------------------------------------
def test_icp_singular_and_identity_on_gpu(golden_dir):
    """utest/ui/icp/GeneralTests.cpp:152-210 through the HIP path."""
    nX = 10
    d = np.float32(0.1)
    pts = np.array([[d * x - 0.5, d * y - 0.5, 0] for x in range(nX) for y in range(nX)], np.float32)
    ref = pts.copy()
    ref[:, 2] = 1.0
    n = np.tile(np.array([0, 0, 1], np.float32), (100, 1))
    kw = dict(max_dist=float("inf"), trim_ratio=1.0, max_normal_angle=None, min_diff_rot=0.001, min_diff_trans=0.01,
              smooth_length=4, max_iters=40, counter_first=True)
    g, o = yaml_pair(**kw)
    T = g(reading=(pts, None), reference=(ref, n))
    o.init_reference(ref, n)
    To = o.compute(pts, None, np.eye(4))
    exp = np.eye(4, dtype=np.float32)
    exp[2, 3] = 1
    assert np.linalg.norm(T - exp) <= 1e-5 * np.linalg.norm(exp)
    assert_pose_close(T, To, 1e-6, 1e-6)
    assert g.stats.iterations == o.stats.iterations
    car = np.load(os.path.join(golden_dir, "car_clouds.npz"))["ref3D"]
    g2, _ = yaml_pair(**kw)
    T2 = g2(reading=(car[:, :3], car[:, 3:6]), reference=(car[:, :3], car[:, 3:6]))
    I = np.eye(4, dtype=np.float32)
    assert np.linalg.norm(T2 - I) <= 1e-4 * np.linalg.norm(I)


def test_valid_t3d_on_gpu(golden_dir):
    """utest/utest.h:65-86 validate3dTransformation on car_cloud401 -> car_cloud400 (tolerance 0.1 m / 0.1 rad)."""
    gz = np.load(os.path.join(golden_dir, "car_clouds.npz"))
    ref, data, valid = gz["ref3D"], gz["data3D"], gz["validT3d"]
    kw = dict(max_dist=float("inf"), trim_ratio=0.85, max_normal_angle=None, min_diff_rot=0.001, min_diff_trans=0.001,
              smooth_length=3, max_iters=40, counter_first=True)
    g, o = yaml_pair(**kw)
    T = g(reading=(data, None), reference=(ref[:, :3], ref[:, 3:6]))
    assert abs(np.linalg.norm(valid[:3, 3]) - np.linalg.norm(T[:3, 3])) < 0.1
    _, ang = orc.pose_error(valid, T)
    assert ang < 0.1
    o.init_reference(ref[:, :3], ref[:, 3:6])
    To = o.compute(data, None, np.eye(4))
    assert_pose_close(T, To)
    assert g.stats.iterations == o.stats.iterations


REF_PINS = [
    ("Matcher.cpp:68-93 maxDist 1.0", 1.0, 0.85, None),
    ("Matcher.cpp:68-93 maxDist 0.5", 0.5, 0.85, None),
    ("Outliers.cpp:49-56 MaxDistOutlierFilter3D 1.0", float("inf"), None, 1.0),
]


@pytest.mark.parametrize("name,max_dist,trim,max_out", REF_PINS, ids=[p[0].split()[0] + f"-{k}" for k, p in enumerate(REF_PINS)])
def test_reference_unit_test_pins_on_gpu(golden_dir, name, max_dist, trim, max_out):
    """utest/ui/Matcher.cpp:68-93 (knn 1; epsilon 0 and 0.2 are the same chain here: the search is exact) and
    utest/ui/Outliers.cpp:49-56 through the HIP path: validate3dTransformation's 0.1 / 0.1 against validT3d, and
    iteration-for-iteration agreement with the oracle."""
    gz = np.load(os.path.join(golden_dir, "car_clouds.npz"))
    ref, data, valid = gz["ref3D"], gz["data3D"], gz["validT3d"]
    kw = dict(max_dist=max_dist, trim_ratio=trim, max_normal_angle=None, max_dist_outlier=max_out, min_diff_rot=0.001,
              min_diff_trans=0.001, smooth_length=3, max_iters=40, counter_first=True)
    for eps in (0.0, 0.2):
        g, o = yaml_pair(epsilon=eps, **kw)
        T = g(reading=(data, None), reference=(ref[:, :3], ref[:, 3:6]))
        assert abs(np.linalg.norm(valid[:3, 3]) - np.linalg.norm(T[:3, 3])) < 0.1
        _, ang = orc.pose_error(valid, T)
        assert ang < 0.1
        o.init_reference(ref[:, :3], ref[:, 3:6])
        To = o.compute(data, None, np.eye(4))
        assert g.stats.iterations == o.stats.iterations
        assert np.array_equal(g.stats.trace_kept, o.trace_kept[:g.stats.iterations])
        if trim is not None:
            assert np.array_equal(g.stats.trace_limit.view(np.uint32), o.trace_limit[:g.stats.iterations].view(np.uint32))
        assert_pose_close(T, To, 1e-5, 1e-5)


CONDITIONING = [
    ("SameBoxNoNoise", 1.0, 0.0, 0.0, True, 1e-6),
    ("SameBoxNoNoiseScale50", 50.0, 0.0, 0.0, True, 1e-4),
    ("DifferentBoxNoNoise", 1.0, 0.0, 0.0, False, 1e-5),
    ("SameBoxNoise", 1.0, 1.0, 30.0, True, 1e-6),
    ("DifferentBoxNoise", 1.0, 0.135, 20.0, False, 1e-5),
]


@pytest.mark.parametrize("name,scale,tstd,rstd,same,eps", CONDITIONING, ids=[c[0] for c in CONDITIONING])
def test_conditioning_on_gpu(name, scale, tstd, rstd, same, eps):
    """utest/ui/icp/Conditioning.cpp:347-470 through the HIP path: the reference's own tolerance contract AND
    agreement with the oracle."""
    cases = syn.conditioning_cases(10000, scale, tstd, rstd, same)
    kw = dict(matcher="MirrorMatcher", max_dist=float("inf"), trim_ratio=None, max_normal_angle=None, min_diff_rot=1e-5,
              min_diff_trans=1e-4, smooth_length=3, max_iters=30)
    g = ICP(IcpConfig(**kw))
    for c in cases:
        o = orc.OracleIcp(orc.OracleConfig(matcher=1, max_dist=float("inf"), trim_ratio=-1, max_normal_angle=-1,
                                           min_diff_rot=1e-5, min_diff_trans=1e-4, smooth_length=3, max_iters=30))
        o.init_reference(c.ref_xyz, c.ref_normals)
        To = o.compute(c.read_xyz, c.read_normals, c.initial_guess)
        assert g.init_reference(c.ref_xyz, c.ref_normals)
        T = g.compute(c.read_xyz, c.read_normals, c.initial_guess)
        dt, ang = orc.pose_error(c.T_origin_read, T)
        assert np.all(np.abs(dt) < eps) and ang < eps, (c.name, dt, ang)
        assert_pose_close(T, To, max(eps, 1e-6) , max(eps, 1e-6))
        assert g.stats.iterations == o.stats.iterations, (c.name, g.stats.iterations, o.stats.iterations)


@pytest.mark.parametrize("variant", ["default", "nosort", "nograph", "cell0.125", "fixed20"])
def test_scan_to_map_c1_matches_oracle(variant):
    """BASELINE configs[0]: 10k-pt scan vs 100k-pt map, point-to-plane, icp.yaml chain (and 20 fixed iterations)."""
    pair = syn.make_scan_pair(10000, 100000, 0.1, seed=0)
    over = {}
    if variant == "nosort":
        over["sort_queries"] = False
    if variant == "nograph":
        over["use_graph"] = False
    if variant == "cell0.125":
        over["grid_cell"] = 0.125
    if variant == "fixed20":
        over.update(use_differential=False, max_iters=20)
    g, o = yaml_pair(**over)
    g.init_reference(pair.map_xyz, pair.map_normals)
    o.init_reference(pair.map_xyz, pair.map_normals)
    T = g.compute(pair.scan_xyz, pair.scan_normals, pair.T_init)
    To = o.compute(pair.scan_xyz, pair.scan_normals, pair.T_init)
    assert g.stats.iterations == o.stats.iterations
    assert g.stats.max_iters_reached == bool(o.stats.max_iters_reached)
    # per-iteration agreement: trim limit is the same ELEMENT, kept sets have the same size, poses track each other
    assert np.array_equal(g.stats.trace_limit.view(np.uint32), o.trace_limit.view(np.uint32))
    assert np.array_equal(g.stats.trace_kept, o.trace_kept)
    for Tg_i, To_i in zip(g.stats.trace_T, o.trace_T):
        assert_pose_close(Tg_i, To_i, 1e-5, 1e-5)
    assert_pose_close(T, To, 1e-5, 1e-5)
    dt, ang = orc.pose_error(pair.T_gt, T)
    assert np.linalg.norm(dt) < 0.02 and ang < 0.005
    assert g.stats.kept_pairs == o.stats.kept_pairs and g.stats.matched_pairs == o.stats.matched_pairs


def test_resident_reading_reuse_and_reinit():
    """Mapper pattern (Mapper.cpp:349-393): one initReference, several compute() calls, then a new reference."""
    pair = syn.make_scan_pair(5000, 50000, 0.1, seed=2)
    g, o = yaml_pair()
    g.init_reference(pair.map_xyz, pair.map_normals)
    o.init_reference(pair.map_xyz, pair.map_normals)
    g.set_reading(pair.scan_xyz, pair.scan_normals)
    Ta = g.compute_resident(pair.T_init)
    Tb = g.compute_resident(pair.T_init)
    assert np.array_equal(Ta, Tb)   # the whole chain is run-independent: integer histograms, fixed-order fp64 sums
    To = o.compute(pair.scan_xyz, pair.scan_normals, pair.T_init)
    assert_pose_close(Ta, To, 1e-5, 1e-5)
    Tc = g.compute_resident(pair.T_gt)
    Toc = o.compute(pair.scan_xyz, pair.scan_normals, pair.T_gt)
    assert_pose_close(Tc, Toc, 1e-5, 1e-5)
    pair2 = syn.make_scan_pair(3000, 30000, 0.1, seed=4)
    g.init_reference(pair2.map_xyz, pair2.map_normals)
    o.init_reference(pair2.map_xyz, pair2.map_normals)
    T2 = g.compute(pair2.scan_xyz, pair2.scan_normals, pair2.T_init)
    To2 = o.compute(pair2.scan_xyz, pair2.scan_normals, pair2.T_init)
    assert_pose_close(T2, To2, 1e-5, 1e-5)


@pytest.mark.parametrize("min_diff,max_iters,smooth,expect", [
    (None, 15, 3, "one chunk"),            # icp.yaml thresholds: done inside the first replay
    (3.0e-5, 15, 3, "several chunks"),     # tighter thresholds: stops by itself after the first chunk
    (0.0, 13, 3, "counter mid-chunk"),     # thresholds that are never met: the Counter ends it at 13 = 2 chunks + 3
    (0.0, 40, 5, "counter after 8 chunks"),
])
def test_chunked_graph_replay_equals_eager_and_oracle(min_diff, max_iters, smooth, expect):
    """A chain that stops by its own checkers replays a captured graph of 5 iterations until `done` (o3s_icp.hip,
    compute_launch / compute_finish).  The first call of a handle runs eagerly, the second captures, the third and
    fourth replay: all four must agree bit for bit — iterations, per-iteration kept counts, limits and poses — and with
    the oracle, whether the chain needs one chunk, several, or runs into the Counter limit in the middle of one."""
    pair = syn.make_scan_pair(6000, 60000, 0.1, seed=12, trans=0.3, rot_deg=5.0)
    kw = dict(max_iters=max_iters, smooth_length=smooth)
    if min_diff is not None:
        kw.update(min_diff_rot=min_diff, min_diff_trans=min_diff)
    g = ICP(IcpConfig(**kw))
    o = orc.OracleIcp(orc.OracleConfig(**kw), threads=8)
    assert g.init_reference(pair.map_xyz, pair.map_normals) and o.init_reference(pair.map_xyz, pair.map_normals) == orc.OK
    g.set_reading(pair.scan_xyz, pair.scan_normals)
    runs = []
    for _ in range(4):
        T = g.compute_resident(pair.T_init)
        n = g.stats.iterations
        runs.append((T.copy(), n, g.stats.trace_kept[:n].copy(), g.stats.trace_limit[:n].copy(), g.stats.max_iters_reached))
    for r in runs[1:]:
        assert r[1] == runs[0][1] and r[4] == runs[0][4]
        assert np.array_equal(r[0], runs[0][0]) and np.array_equal(r[2], runs[0][2]) and np.array_equal(r[3], runs[0][3])
    To = o.compute(pair.scan_xyz, pair.scan_normals, pair.T_init)
    n = runs[0][1]
    assert n == o.stats.iterations and np.array_equal(runs[0][2], o.trace_kept[:n])
    assert_pose_close(runs[0][0], To, 1e-5, 1e-5)
    if expect == "one chunk":
        assert n <= 5 and not runs[0][4]
    elif expect == "several chunks":
        assert 5 < n < max_iters and not runs[0][4]
    else:
        assert n == max_iters and runs[0][4]


def test_graph_is_dropped_when_a_captured_buffer_moves():
    """One handle, readings of size N1, N1 (captures the graph), N2 just beyond the 1/8 growth slack of the candidate
    segments (d_cand reallocates, the keyed buffers need not), then N1 again: the replay of the old graph would read freed
    memory.  Every allocation of the handle bumps a generation that is part of the graph key (o3s_icp.hip, DevBuf::gen),
    so the fourth call re-captures; all N1 runs must agree bit for bit and with the oracle."""
    n1 = 8000
    n2 = int(1.125 * n1) + 6
    pair = syn.make_scan_pair(n2, 60000, 0.1, seed=21)
    kw = dict(use_differential=False, max_iters=8)
    g = ICP(IcpConfig(**kw))
    o = orc.OracleIcp(orc.OracleConfig(**kw), threads=8)
    assert g.init_reference(pair.map_xyz, pair.map_normals) and o.init_reference(pair.map_xyz, pair.map_normals) == orc.OK
    a_xyz, a_n = pair.scan_xyz[:n1], pair.scan_normals[:n1]
    runs = []
    for xyz, nn in [(a_xyz, a_n), (a_xyz, a_n), (a_xyz, a_n), (pair.scan_xyz, pair.scan_normals), (a_xyz, a_n), (a_xyz, a_n), (a_xyz, a_n)]:
        T = g.compute(xyz, nn, pair.T_init)
        runs.append((len(xyz), T.copy(), g.stats.trace_kept.copy(), g.stats.trace_limit.copy()))
    first = runs[0]
    for r in runs:
        if r[0] == n1:
            assert np.array_equal(r[1], first[1]) and np.array_equal(r[2], first[2]) and np.array_equal(r[3], first[3])
    To = o.compute(a_xyz, a_n, pair.T_init)
    assert np.array_equal(first[2], o.trace_kept[:8]) and np.array_equal(first[3].view(np.uint32), o.trace_limit[:8].view(np.uint32))
    assert_pose_close(first[1], To, 1e-5, 1e-5)
    To2 = o.compute(pair.scan_xyz, pair.scan_normals, pair.T_init)
    assert_pose_close(runs[3][1], To2, 1e-5, 1e-5)


def test_error_mapping_on_gpu():
    rng = np.random.default_rng(3)
    ref = rng.uniform(-1, 1, (500, 3)).astype(np.float32)
    n = np.tile(np.array([0, 0, 1], np.float32), (500, 1))
    g = ICP(IcpConfig())
    with pytest.raises(RuntimeError):
        g.compute(ref, n, np.eye(4))                       # not initialised
    assert g.init_reference(np.zeros((0, 3), np.float32), None) is False   # ICP.cpp:295-298
    assert g.init_reference(ref, n)
    with pytest.raises(RuntimeError):
        g.compute(np.zeros((0, 3), np.float32), None, np.eye(4))   # ICP.cpp:357-359
    with pytest.raises(ConvergenceError):
        g.compute(ref + 100.0, n, np.eye(4))               # no matches within maxDist -> Matches.cpp:76-77
    bad = np.eye(4)
    bad[:3, :3] *= 1.1
    with pytest.raises(TransformationError):
        g.compute(ref, n, bad)                             # TransformationsImpl.cpp:73-74
    # the reference's own non-orthogonal matrix (utest/ui/Transformations.cpp:133-147): |1 - det| = 1.98e-3 against the 1e-3 band
    from ref_pins import REF_T3D_NOT_RIGID
    with pytest.raises(TransformationError):
        g.compute(ref, n, REF_T3D_NOT_RIGID)
    assert orc.OracleIcp(orc.OracleConfig()).init_reference(ref, n) == orc.OK
    g2 = ICP(IcpConfig(trim_ratio=None, max_normal_angle=None))
    g2.init_reference(ref, n)
    with pytest.raises(ConvergenceError):
        g2.compute(ref + 100.0, n, np.eye(4))              # ErrorMinimizer.cpp:75-77
    g3 = ICP(IcpConfig())
    g3.init_reference(ref, None)
    with pytest.raises(RuntimeError):
        g3.compute(ref, n, np.eye(4))                      # point-to-plane without reference normals
    # after errors the handle is still usable
    T = g.compute(ref, n, np.eye(4))
    assert np.allclose(T, np.eye(4), atol=1e-4)


def test_ragged_sizes():
    """Sizes that are not multiples of the wave / block / XCD group, and tiny clouds."""
    for N, M in [(1, 1), (63, 65), (257, 1000), (2049, 4099)]:
        rng = np.random.default_rng(N)
        ref = rng.uniform(-1, 1, (M, 3)).astype(np.float32)
        g, o = both(dict(max_dist=float("inf")), dict(max_dist=float("inf")))
        g.init_reference(ref, np.zeros_like(ref))
        o.init_reference(ref, np.zeros_like(ref))
        q = rng.uniform(-1, 1, (N, 3)).astype(np.float32)
        a = g.find_closests(q)
        b = o.find_closests(q, brute=True)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_compute_batch_matches_sequential():
    """BASELINE configs[2] (scaled down): independent scan/submap pairs issued together must give exactly what each pair
    gives on its own, including a failing pair (status reported per pair, Mapper.cpp:420-422 keeps the prior)."""
    from open3d_slam_advanced_rss_2024_public_amd import compute_batch, parallel

    pairs = [syn.make_scan_pair(4000, 30000, 0.1, seed=20 + k) for k in range(6)]
    icps, solo = [], []
    for k, p in enumerate(pairs):
        icp = ICP(IcpConfig())
        icp.init_reference(p.map_xyz, p.map_normals)
        scan = p.scan_xyz + (400.0 if k == 4 else 0.0)
        icp.set_reading(scan, p.scan_normals)
        icps.append(icp)
        one = ICP(IcpConfig())
        one.init_reference(p.map_xyz, p.map_normals)
        try:
            solo.append((one.compute(scan, p.scan_normals, p.T_init), 0, one.stats.iterations))
        except ConvergenceError:
            solo.append((None, 5, one.stats.iterations))
    poses, codes, stats = compute_batch(icps, [p.T_init for p in pairs])
    for k in range(6):
        assert codes[k] == solo[k][1]
        if codes[k] == 0:
            assert np.array_equal(poses[k], solo[k][0])
            assert stats[k].iterations == solo[k][2]
        else:
            assert poses[k] is None
    # the sharding front-end in single-process mode uses the same runner
    dicts = [dict(map_xyz=p.map_xyz, map_normals=p.map_normals, scan_xyz=p.scan_xyz, scan_normals=p.scan_normals, T_init=p.T_init)
             for p in pairs[:3]]
    res = parallel.run_pairs_sharded(dicts, parallel.gpu_runner(IcpConfig(), 0))
    for k in range(3):
        assert res[k][1] == 0 and np.array_equal(res[k][0], solo[k][0])


def test_cpp_shim_runs_the_same_registration(tmp_path):
    """The drop-in boundary from C++: a g++-built program that only includes cpp/o3s_icp.hpp and links the .so gives
    the same pose as the Python mirror (same C ABI underneath), maps an empty reading to std::runtime_error, and the
    SubmapHip path (insert -> patch -> reference on the device) registers against the same map."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "open3d_slam_advanced_rss_2024_public_amd")
    exe = tmp_path / "shim_roundtrip"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(root, "include"), "-I" + os.path.join(pkg, "cpp"),
                           os.path.join(root, "tests", "cpp", "shim_roundtrip.cpp"), "-L" + pkg, "-lo3dslam_icp_hip", "-Wl,-rpath," + pkg,
                           "-o", str(exe)])
    sp = syn.make_scan_pair(6000, 50000, 0.1, seed=8)
    from open3d_slam_advanced_rss_2024_public_amd.icp import as_xyzw

    files = {}
    for name, arr in (("ref", as_xyzw(sp.map_xyz)), ("refn", sp.map_normals.astype(np.float32)), ("scan", as_xyzw(sp.scan_xyz)),
                      ("scann", sp.scan_normals.astype(np.float32)), ("T0", np.ascontiguousarray(sp.T_init.astype(np.float32).T))):
        files[name] = str(tmp_path / f"{name}.f32")
        np.ascontiguousarray(arr, np.float32).tofile(files[name])
    out = subprocess.run([str(exe), files["ref"], files["refn"], str(sp.map_xyz.shape[0]), files["scan"], files["scann"],
                          str(sp.scan_xyz.shape[0]), files["T0"]], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr)
    lines = dict(l.split(" ", 1) for l in out.stdout.strip().splitlines())
    T_cpp = np.array(lines["T"].split(), np.float32).reshape(4, 4).T
    icp = ICP(IcpConfig())
    assert icp.init_reference(sp.map_xyz, sp.map_normals)
    T_py = icp.compute(sp.scan_xyz, sp.scan_normals, sp.T_init)
    assert np.array_equal(T_cpp, T_py) and int(lines["iters"]) == icp.stats.iterations
    assert lines["empty"] == "runtime_error"
    assert int(lines["patch"]) == sp.map_xyz.shape[0]
    T2 = np.array(lines["T2"].split(), np.float32).reshape(4, 4).T
    dt, ang = orc.pose_error(T_py, T2)          # the map went through fp64 (x - 0.5) + 0.5: equal up to that rounding
    assert np.linalg.norm(dt) <= 1e-5 and ang <= 1e-5
    # DenseMapHip: voxel count, carved voxels and survivors agree with the oracle's dense map on the same cloud
    mp = as_xyzw(sp.map_xyz)[:, :3].astype(np.float64)
    mp[:, 0] -= 0.5
    om = orc.DenseMap(0.1)
    om.insert(mp, sp.map_normals.astype(np.float32).astype(np.float64))
    v0 = om.size()
    removed = om.carve(np.array([[50.0, 0.013, 0.017]]), [0.0, 0.0, 0.0], 0.1, 20.0, 0.1)
    assert [int(v) for v in lines["dense"].split()] == [v0, removed, om.size(), 1] and removed > 0


def test_device_resident_inputs_handed_over_with_an_event_equal_the_host_path():
    """o3s_icp_wait_event + init_reference_dev_async + set_reading_dev: a producer stream (torch's) fills the reference and the
    reading in HBM, records an event, and the handle orders its own stream behind it — no host wait, same bits as the host
    path.  Runs IN this process, after every other test of the module has used the library: torch is imported here, late.
    (Round 2 moved it to a child process because torch then failed to initialise; the cause was two ROCm runtimes in one
    process — the library had bound to /opt/rocm's by soname, torch mapped its own beside it.  _lib.lib() now maps torch's
    runtime first, tests/test_abi.py pins that, and this test is back where it belongs.)"""
    import torch

    from open3d_slam_advanced_rss_2024_public_amd import _lib

    kinds = [os.path.basename(p).split(".so")[0] for p in _lib.loaded_rocm_runtimes()]
    assert len(kinds) == len(set(kinds)), _lib.loaded_rocm_runtimes()
    sp = syn.make_scan_pair(6000, 50000, 0.1, seed=12)
    dev = torch.device("cuda", 0)
    producer = torch.cuda.Stream(dev)
    with torch.cuda.stream(producer):
        ref = torch.ones((sp.map_xyz.shape[0], 4), dtype=torch.float32, device=dev)
        ref[:, :3] = torch.from_numpy(np.ascontiguousarray(sp.map_xyz, np.float32)).to(dev, non_blocking=True)
        refn = torch.from_numpy(np.ascontiguousarray(sp.map_normals, np.float32)).to(dev, non_blocking=True).contiguous()
        rd = torch.ones((sp.scan_xyz.shape[0], 4), dtype=torch.float32, device=dev)
        rd[:, :3] = torch.from_numpy(np.ascontiguousarray(sp.scan_xyz, np.float32)).to(dev, non_blocking=True)
        rdn = torch.from_numpy(np.ascontiguousarray(sp.scan_normals, np.float32)).to(dev, non_blocking=True).contiguous()
        ev = torch.cuda.Event()
        ev.record(producer)
    g = ICP(IcpConfig())
    g.wait_event(ev.cuda_event)
    assert g.init_reference_dev_async(ref.data_ptr(), refn.data_ptr(), ref.shape[0])
    g.set_reading_dev(rd.data_ptr(), rdn.data_ptr(), rd.shape[0])
    T_dev = g.compute_resident(sp.T_init)
    host = ICP(IcpConfig())
    assert host.init_reference(sp.map_xyz, sp.map_normals)
    T_host = host.compute(sp.scan_xyz, sp.scan_normals, sp.T_init)
    assert np.array_equal(T_dev, T_host)
    n = host.stats.iterations
    assert g.stats.iterations == n
    assert np.array_equal(g.stats.trace_limit[:n].view(np.uint32), host.stats.trace_limit[:n].view(np.uint32))
    assert np.array_equal(g.stats.trace_kept[:n], host.stats.trace_kept[:n])


def test_fused_selection_kernel_gives_the_bits_of_the_two_kernel_chain(monkeypatch, hooks_lib):
    """Up to 131 k points k_sel_finish + k_normal_eq run as one launch (k_sel_ne) — except inside o3s_icp_compute_batch.  A
    pair must not depend on how it was issued: same limits, same kept counts, same pose bits, eager and replayed."""
    sp = syn.make_scan_pair(30000, 200000, 0.1, seed=31)
    out = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("O3S_FUSE", fuse)
        g = ICP(IcpConfig(use_differential=False, max_iters=12))
        assert g.init_reference(sp.map_xyz, sp.map_normals)
        g.set_reading(sp.scan_xyz, sp.scan_normals)
        Ts = [g.compute_resident(sp.T_init) for _ in range(3)]      # the third call replays a captured graph
        assert all(np.array_equal(Ts[0], T) for T in Ts[1:])
        out[fuse] = (Ts[0], g.stats.trace_limit.copy(), g.stats.trace_kept.copy(), g.stats.trace_T.copy())
    for a, b in zip(out["1"], out["0"]):
        assert np.array_equal(a, b)


def test_compute_in_two_halves_gives_the_bits_of_the_one_call():
    """o3s_icp_compute_resident_launch / _finish: the chain goes onto the stream in the first half without anybody looking at its
    result, the second half waits, issues what an eagerly issued chain still needs, and composes the pose.  Where the host looks
    never decides what the chain computes: per-iteration limits, kept counts, every T_iter and the pose equal the one-call form —
    eagerly issued chains (a fresh handle guesses two iterations and has to go on; a warm one guesses right), captured and replayed
    ones, the icp.yaml chain that stops by itself and a fixed count, and a chain that fails (nothing within maxDist)."""
    sp = syn.make_scan_pair(20000, 150000, 0.1, seed=41)
    for kw in (dict(use_differential=True, max_iters=15), dict(use_differential=False, max_iters=7), dict(use_differential=True, max_iters=15, use_graph=False)):
        one, two = ICP(IcpConfig(**kw)), ICP(IcpConfig(**kw))
        for g in (one, two):
            assert g.init_reference(sp.map_xyz, sp.map_normals)
            g.set_reading(sp.scan_xyz, sp.scan_normals)
        for call in range(4):                     # eager with a cold hint, eager with a warm one / captured, replayed, replayed
            Ta = one.compute_resident(sp.T_init)
            two.compute_resident_launch(sp.T_init)
            Tb = two.compute_resident_finish()
            assert np.array_equal(Ta, Tb), (kw, call)
            assert one.stats.iterations == two.stats.iterations
            n = one.stats.iterations
            assert np.array_equal(one.stats.trace_limit[:n].view(np.uint32), two.stats.trace_limit[:n].view(np.uint32))
            assert np.array_equal(one.stats.trace_kept[:n], two.stats.trace_kept[:n]) and np.array_equal(one.stats.trace_T[:n], two.stats.trace_T[:n])
        # another reading size on the same handles: new shapes, issued eagerly again
        half = len(sp.scan_xyz) // 2 + 7
        for g in (one, two):
            g.set_reading(sp.scan_xyz[:half], sp.scan_normals[:half])
        Ta = one.compute_resident(sp.T_init)
        two.compute_resident_launch(sp.T_init)
        assert np.array_equal(Ta, two.compute_resident_finish())
    far = ICP(IcpConfig())
    assert far.init_reference(sp.map_xyz, sp.map_normals)
    far.set_reading(sp.scan_xyz + np.float32(500.0), sp.scan_normals)
    far.compute_resident_launch(sp.T_init)
    with pytest.raises(Exception):
        far.compute_resident_finish()
    with pytest.raises(Exception):                 # a second finish without a launch
        far.compute_resident_finish()


def test_first_iteration_index_of_dense_maps_changes_no_bit(monkeypatch, hooks_lib):
    """A map dense enough for the matcher to shrink its cell gets a second index of the same points on a coarser grid, searched by
    iteration 0 of every chain (no incumbents yet: csrc/o3s_icp.hip init_reference_impl step 4).  The search is exact on either
    grid, so nothing may move: per-iteration limits and kept counts, every T_iter, the pose — with the index off
    (O3S_FIRST_GRID=0), with the library's edge and with an odd one; eager, captured and replayed; the icp.yaml chain and a fixed
    count; and the oracle agrees with all of them."""
    sp = syn.make_scan_pair(40_000, 1_500_000, 0.02, seed=23, radius=5.0)
    for kw in (dict(use_differential=False, max_iters=8), dict(use_differential=True, max_iters=15)):
        out = {}
        for edge in ("0", None, "0.061"):
            if edge is None:
                monkeypatch.delenv("O3S_FIRST_GRID", raising=False)
            else:
                monkeypatch.setenv("O3S_FIRST_GRID", edge)
            g = ICP(IcpConfig(**kw))
            assert g.init_reference(sp.map_xyz, sp.map_normals)
            g.set_reading(sp.scan_xyz, sp.scan_normals)
            Ts = [g.compute_resident(sp.T_init) for _ in range(3)]      # eager, captured, replayed
            assert all(np.array_equal(Ts[0], T) for T in Ts[1:]), edge
            out[edge] = (Ts[0], g.stats.trace_limit.copy(), g.stats.trace_kept.copy(), g.stats.trace_T.copy(), int(g.stats.iterations))
        for edge in (None, "0.061"):
            for a, b in zip(out["0"], out[edge]):
                assert np.array_equal(a, b), (kw, edge)
        o = orc.OracleIcp(orc.OracleConfig(**kw), threads=8)
        o.init_reference(sp.map_xyz, sp.map_normals)
        To, _ = o.compute(sp.scan_xyz, sp.scan_normals, sp.T_init, raise_on_error=False)
        dt, ang = orc.pose_error(To, out[None][0])
        assert np.linalg.norm(dt) < 1e-6 and ang < 1e-6 and o.stats.iterations == out[None][4]
    monkeypatch.delenv("O3S_FIRST_GRID", raising=False)


def test_reading_sort_is_stable_whatever_the_arrival_order_of_its_atomic(monkeypatch, hooks_lib):
    """The counting sort that puts the reading into grid order hands out slots with an integer atomic; the place of a point
    INSIDE its bin must be its input rank all the same (k_read_place), or the order of every fp64 sum downstream would be
    the scheduler's.  O3S_SCATTER_ORDER=1 deals the points to the scatter's threads back to front, which reverses the
    arrival order: the processing order, every per-iteration limit / kept count and every pose bit must not move — on C1,
    on a reading with many points per bin (coarse bins) and on one piled up in a single border bin (far outside the map)."""
    c1 = syn.make_scan_pair(10000, 100000, 0.1, seed=7)
    dense = syn.make_scan_pair(40000, 60000, 0.1, seed=8)
    far_xyz = c1.scan_xyz.copy()
    far_xyz[: len(far_xyz) // 3] += np.array([60.0, 0.0, 0.0], np.float32)  # a third of the scan beyond the grid: ONE bin
    cases = [("c1", c1.scan_xyz, c1.scan_normals, c1, {}),
             ("coarse bins", dense.scan_xyz, dense.scan_normals, dense, dict(grid_cell=2.0)),
             ("one border bin", far_xyz, c1.scan_normals, c1, {})]
    for name, sxyz, sn, pair, over in cases:
        out = {}
        for order in ("0", "1"):
            monkeypatch.setenv("O3S_SCATTER_ORDER", order)
            g = ICP(IcpConfig(use_differential=False, max_iters=10, use_graph=False, **over))
            assert g.init_reference(pair.map_xyz, pair.map_normals)
            T = g.compute(sxyz, sn, pair.T_init)
            perm = g.reading_order(len(sxyz))
            out[order] = (perm, T, g.stats.trace_limit.copy(), g.stats.trace_kept.copy(), g.stats.trace_T.copy())
        perm = out["0"][0]
        assert len(perm) == len(sxyz) and np.array_equal(np.sort(perm), np.arange(len(sxyz))), name
        for a, b in zip(out["0"], out["1"]):
            assert np.array_equal(a, b), name
    monkeypatch.delenv("O3S_SCATTER_ORDER")


def test_multi_block_selection_sweep_of_large_readings_equals_the_single_block_one(monkeypatch, hooks_lib):
    """Readings with more classify blocks than the finishing block has threads (> 262 k points: C4) sweep their trim candidates
    on many blocks (k_sel_partial) before k_sel_finish ranks the few undecided ones; O3S_SEL_PARTIAL=0 keeps the single-block
    sweep.  The limit is the same ELEMENT and the kept sets have the same size in every iteration (integer work); the poses
    agree to 1e-6 (the fp64 sums are folded in another fixed order before their one rounding), eager and replayed."""
    sp = syn.make_scan_pair(300_000, 700_000, 0.05, seed=41)
    out = {}
    for part in ("1", "0"):
        monkeypatch.setenv("O3S_SEL_PARTIAL", part)
        g = ICP(IcpConfig(use_differential=False, max_iters=8))
        assert g.init_reference(sp.map_xyz, sp.map_normals)
        g.set_reading(sp.scan_xyz, sp.scan_normals)
        Ts = [g.compute_resident(sp.T_init) for _ in range(3)]      # the third call replays a captured graph
        assert all(np.array_equal(Ts[0], T) for T in Ts[1:])
        out[part] = (Ts[0], g.stats.trace_limit.copy(), g.stats.trace_kept.copy())
    assert np.array_equal(out["1"][1].view(np.uint32), out["0"][1].view(np.uint32))
    assert np.array_equal(out["1"][2], out["0"][2])
    assert np.abs(out["1"][0] - out["0"][0]).max() <= 1e-6
    dt, ang = orc.pose_error(sp.T_gt, out["1"][0])
    assert np.linalg.norm(dt) < 0.01 and ang < 0.003
