"""Oracle pins for the open3d_slam side of the path: voxel index / hash / voxelise / crop / conversion."""
import math

import numpy as np

from oracle import oracle as orc


def test_voxel_idx_reciprocal_form():
    """getVoxelIdx(p, InverseVoxelSize) (VoxelHashMap.hpp:37-51): int(floor(p * (1.0/voxel))) in fp64."""
    pts = np.array([[-0.05, 0.3, 0.0], [0.1, -0.1, 0.29999999999999999], [1e-300, -1e-300, 7.0]], np.float64)
    idx = orc.voxel_idx(pts, 0.1)
    inv = 1.0 / 0.1
    exp = np.floor(pts * inv).astype(np.int32)
    assert np.array_equal(idx, exp)
    assert idx[0, 0] == -1            # p=-0.05, v=0.1 => -1
    assert idx[0, 1] == int(math.floor(0.3 * (1.0 / 0.1)))
    assert idx[2, 1] == -1 and idx[2, 0] == 0
    # the dividing overloads (VoxelHashMap.hpp:53-61) can differ by one cell from the reciprocal form
    rng = np.random.default_rng(0)
    k = rng.integers(-1000, 1000, (20000, 3))
    p = k * 0.1  # points exactly on nominal boundaries
    a = orc.voxel_idx(p, 0.1)
    b = orc.voxel_idx_div(p, 0.1)
    assert np.array_equal(b, np.floor(p / 0.1).astype(np.int32))
    assert np.any(a != b)  # documents gotcha (v) of SURVEY Appendix A


def test_voxel_hash_wraps_and_truncates():
    """EigenVec3iHash (VoxelHashMap.hpp:25-35): size_t arithmetic (negatives wrap mod 2^64) truncated to 32 bits."""
    idx = np.array([[1, 2, 3], [-1, 0, 0], [0, -1, 0], [-5, -7, -9], [2**20, -2**20, 17]], np.int32)
    h = orc.voxel_hash(idx)
    sl = 17191
    for row, hv in zip(idx, h):
        v = (int(row[0]) + int(row[1]) * sl + int(row[2]) * sl * sl) % (1 << 64)
        assert int(hv) == v % (1 << 32)


def test_crop_predicates():
    """croppers.cpp:121-167 predicates + invert flag (croppers.cpp:57-59)."""
    rng = np.random.default_rng(1)
    p = rng.uniform(-20, 20, (5000, 3))
    c = (1.0, -2.0, 0.5)
    d = np.linalg.norm(p - np.array(c), axis=1)
    assert np.array_equal(orc.crop_mask(orc.make_cropper("MaxRadius", 15.0, centre=c), p), d <= 15.0)
    assert np.array_equal(orc.crop_mask(orc.make_cropper("MinRadius", 5.0, centre=c), p), d >= 5.0)
    assert np.array_equal(orc.crop_mask(orc.make_cropper("MinMaxRadius", 5.0, 15.0, centre=c), p), (d >= 5) & (d <= 15))
    dxy = np.linalg.norm((p - np.array(c))[:, :2], axis=1)
    cyl = (p[:, 2] >= -3) & (p[:, 2] <= 4) & (dxy <= 10)
    assert np.array_equal(orc.crop_mask(orc.make_cropper("Cylinder", 10.0, -3.0, 4.0, centre=c), p), cyl)
    assert np.array_equal(orc.crop_mask(orc.make_cropper("MaxRadius", 15.0, centre=c, invert=True), p), ~(d <= 15.0))


def test_voxelize_within_crop():
    """voxelizeWithinCroppingVolume (helpers.cpp:117-192): pass-through first, then per-voxel mean + normalised normal."""
    rng = np.random.default_rng(2)
    p = rng.uniform(-3, 3, (4000, 3))
    n = rng.normal(size=(4000, 3))
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    n[5] = np.nan  # NaN normals are skipped in the sum but the point still counts (helpers.cpp:34-38)
    crop = orc.make_cropper("MaxRadius", 2.0)
    op, on, oi = orc.voxelize_within_crop(crop, 0.5, p, n)
    inside = np.linalg.norm(p, axis=1) <= 2.0
    n_pass = int((~inside).sum())
    assert np.array_equal(op[:n_pass], p[~inside])
    assert np.all(oi[:n_pass] == np.iinfo(np.int32).min)
    keys = np.floor(p[inside] * (1.0 / 0.5)).astype(np.int32)
    uk = np.unique(keys, axis=0)
    assert len(op) == n_pass + len(uk)
    got = {tuple(k): (pp, nn) for k, pp, nn in zip(oi[n_pass:], op[n_pass:], on[n_pass:])}
    pin, nin = p[inside], n[inside]
    for k in uk:
        sel = np.all(keys == k, axis=1)
        mp = pin[sel].sum(axis=0) / sel.sum()
        nn = nin[sel]
        good = ~np.isnan(nn).any(axis=1)
        mn = nn[good].sum(axis=0) / sel.sum()
        mn = mn / np.linalg.norm(mn)
        gp, gn = got[tuple(k)]
        assert np.allclose(gp, mp, rtol=0, atol=1e-12)
        assert np.allclose(gn, mn, rtol=0, atol=1e-12)


def test_o3d_voxel_downsample_and_conversion():
    rng = np.random.default_rng(3)
    p = rng.uniform(-5, 5, (3000, 3))
    n = rng.normal(size=(3000, 3))
    op, on, oi = orc.voxel_downsample_o3d(0.7, p, n)
    mn = p.min(axis=0) - 0.35
    keys = np.floor((p - mn) / 0.7).astype(np.int32)
    uk = np.unique(keys, axis=0)
    assert len(op) == len(uk)
    got = {tuple(k): pp for k, pp in zip(oi, op)}
    for k in uk[:50]:
        sel = np.all(keys == k, axis=1)
        assert np.allclose(got[tuple(k)], p[sel].mean(axis=0), atol=1e-12)
    # open3dToPointmatcher (open3d_conversions.cpp:57-118): fp64 -> fp32 round-to-nearest, pad = 1
    xyzw, nn = orc.o3d_to_pm(p, n)
    assert np.array_equal(xyzw[:, :3], p.astype(np.float32)) and np.all(xyzw[:, 3] == 1)
    assert np.array_equal(nn, n.astype(np.float32))


def test_transform_cloud_matches_helpers_cpp():
    """o3d_slam::transform (helpers.cpp:283-318): p' = R p + t exactly for a w = 1 pose; normals rotate only; an
    (almost-)identity pose returns the input followed by the transformed points (helpers.cpp:285-288)."""
    from oracle import oracle as orc

    p = np.array([[1.0, 2.0, 3.0], [-0.5, 0.25, 8.0]])
    n = np.array([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0]])
    T = np.eye(4)
    T[:3, :3] = [[0, -1, 0], [1, 0, 0], [0, 0, 1]]   # +90 deg about z
    T[:3, 3] = [10.0, 20.0, 30.0]
    tp, tn = orc.transform_cloud(T, p, n)
    assert np.array_equal(tp, [[8.0, 21.0, 33.0], [9.75, 19.5, 38.0]])
    assert np.array_equal(tn, [[0.0, 0.0, 1.0], [0.0, 1.0, 0.0]])
    tp, tn = orc.transform_cloud(np.eye(4), p, n)
    assert tp.shape == (4, 3) and np.array_equal(tp[:2], p) and np.array_equal(tp[2:], p) and np.array_equal(tn[2:], n)
    Te = np.eye(4)
    Te[1, 3] = 2e-4                                   # beyond the 1e-4 identity test: no doubling
    assert orc.transform_cloud(Te, p, None)[0].shape == (2, 3)


def test_estimate_normals_known_answers():
    """Oracle restatement of Open3D EstimateNormals(Hybrid) + NormalizeNormals + OrientNormalsTowardsCameraLocation:
    analytic planes, the closed-form eigen solver against numpy's, the < 3 neighbours rule, the strict radius cut."""
    from oracle import oracle as orc

    rng = np.random.default_rng(0)
    xy = rng.uniform(-2, 2, (1500, 2))
    z = 0.3 * xy[:, 0] + 0.1 * xy[:, 1] + 2.0 + rng.normal(0, 1e-3, 1500)
    p = np.c_[xy, z]
    n, nn = orc.estimate_normals(p, 0.5, 10, want_neighbours=True)
    ref = np.array([0.3, 0.1, -1.0]) / np.linalg.norm([0.3, 0.1, -1.0])
    assert np.abs(np.abs(n @ ref) - 1).max() < 5e-3
    assert ((n * (-p)).sum(1) >= 0).all() and np.abs(np.linalg.norm(n, axis=1) - 1).max() < 1e-12
    assert (nn[:, 0] == np.arange(1500)).all()                 # the query is its own nearest neighbour
    for i in (0, 7, 100):                                      # closed-form eigenvector == LAPACK's
        idx = nn[i][nn[i] >= 0]
        w, v = np.linalg.eigh(np.cov(p[idx].T, bias=True))
        assert abs(abs(v[:, 0] @ n[i]) - 1) < 1e-9
    # ascending distance, ties to the lower index, strict d2 < r^2 cut (KDTreeFlann::SearchHybrid)
    q = np.array([[0.0, 0, 1], [0.3, 0, 1], [-0.3, 0, 1], [0, 0.5, 1], [0, 0, 3]])
    n, nn = orc.estimate_normals(q, 0.5, 4, want_neighbours=True)
    assert nn[0].tolist() == [0, 1, 2, -1]                     # 0.5 away is NOT inside the radius
    assert np.array_equal(n[4], [0.0, 0.0, -1.0])             # alone: (0,0,1), flipped towards the origin


def test_carve_known_answer():
    """getIdxsOfCarvedPoints restated: points hanging in the free space the rays cross are carved, the surface the rays
    end on is not (truncation), points outside the subset or with a normal perpendicular to the ray are not."""
    from oracle import oracle as orc

    yy, zz = np.meshgrid(np.linspace(-1, 1, 21), np.linspace(0, 2, 21))
    wall = np.c_[np.full(yy.size, 5.0), yy.ravel(), zz.ravel()]
    blob = np.array([[2.5, 0.0, 1.0], [2.52, 0.01, 1.02], [2.5, 3.0, 1.0], [2.51, 0.0, 1.0]])
    mp = np.r_[wall, blob]
    mn = np.r_[np.tile([-1.0, 0, 0], (len(wall), 1)), [[-1.0, 0, 0], [-2.0, 0, 0], [-1.0, 0, 0], [0.0, 1.0, 0]]]
    sensor = np.array([0.0, 0.0, 1.0])
    rm = orc.carve(wall, mp, mn, sensor, 0.1, 20.0, 0.1, 0.5)
    assert rm[: len(wall)].sum() == 0
    assert rm[len(wall):].tolist() == [True, True, False, False]   # on the ray / off the ray / perpendicular normal
    assert orc.carve(wall, mp, None, sensor, 0.1, 20.0, 0.1, 0.5)[len(wall):].tolist() == [True, True, False, True]
    subset = np.ones(len(mp), bool)
    subset[len(wall)] = False
    assert orc.carve(wall, mp, mn, sensor, 0.1, 20.0, 0.1, 0.5, subset=subset)[len(wall):].tolist() == [False, True, False, False]
    assert orc.carve(wall, mp, mn, sensor, 0.1, 2.0, 0.1, 0.5).sum() == 0   # rays cut at 2 m never reach the blob


def test_overlap_indices_hand_derived():
    """computeIndicesOfOverlappingPoints (open3d_slam/src/helpers.cpp:319-345) on a case small enough to do by hand: 1 m
    voxels, the source shifted by +1 m in x by sourceToTarget."""
    tgt = np.array([[0.2, 0.2, 0.2], [0.7, 0.1, 0.3], [1.5, 0.5, 0.5], [5.5, 5.5, 5.5], [-0.5, 0.5, 0.5]])
    src = np.array([[-0.6, 0.4, 0.4],    # -> (0.4, ..): voxel (0,0,0), which holds targets 0 and 1
                    [0.6, 0.6, 0.6],     # -> (1.6, ..): voxel (1,0,0), target 2
                    [3.0, 3.0, 3.0],     # -> (4.0, 3, 3): voxel (4,3,3), no target
                    [-1.2, 0.9, 0.1]])   # -> (-0.2, ..): voxel (-1,0,0), target 4 (floor of a negative coordinate)
    T = np.eye(4)
    T[0, 3] = 1.0
    i_s, i_t = orc.overlap_indices(src, tgt, T, 1.0, 1)
    assert i_s.tolist() == [0, 1, 3] and i_t.tolist() == [0, 1, 2, 4]
    i_s, i_t = orc.overlap_indices(src, tgt, T, 1.0, 2)       # only voxels with two points of EACH layer: none (source has one per voxel)
    assert i_s.tolist() == [] and i_t.tolist() == []
    src2 = np.vstack([src, [[-0.9, 0.1, 0.1]]])               # a second source point in voxel (0,0,0)
    i_s, i_t = orc.overlap_indices(src2, tgt, T, 1.0, 2)
    assert i_s.tolist() == [0, 4] and i_t.tolist() == [0, 1]


def test_voxelize_attrs_hand_derived():
    """Colour / covariance lanes of voxelizeWithinCroppingVolume (helpers.cpp:30-64, 141-187): last colour of the voxel, mean
    covariance; Open3D's VoxelDownSample averages both; pass-through points come first and keep theirs."""
    pts = np.array([[0.1, 0.1, 0.1], [9.0, 9.0, 9.0], [0.2, 0.3, 0.1], [0.9, 0.9, 0.9], [0.3, 0.2, 0.2]])
    col = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0, 1, 0], [7, -3, 2], [0, 0, 1]], float)   # [7,-3,2] is "invalid" yet kept
    cov = np.stack([np.eye(3).reshape(9) * (k + 1) for k in range(5)])
    c = orc.make_cropper("MaxRadius", 5.0)
    op, _, oi = orc.voxelize_within_crop(c, 0.5, pts)
    oc, ov = orc.voxelize_attrs(0, c, 0.5, pts, col, cov)
    # point 1 is outside the 5 m radius -> passes through first; voxel (0,0,0) holds points 0, 2, 4; voxel (1,1,1) holds point 3
    assert op.shape[0] == 3 and oi[0, 0] == np.iinfo(np.int32).min
    assert oc[0].tolist() == [0.5, 0.5, 0.5] and np.array_equal(ov[0], cov[1])
    assert oc[1].tolist() == [0, 0, 1] and np.allclose(ov[1], np.eye(3).reshape(9) * (1 + 3 + 5) / 3)   # LAST colour, MEAN covariance
    assert oc[2].tolist() == [7, -3, 2] and np.array_equal(ov[2], cov[3])
    oc, ov = orc.voxelize_attrs(1, None, 100.0, pts, col, cov)                                        # one Open3D voxel: means
    assert np.allclose(oc[0], col.mean(axis=0)) and np.allclose(ov[0], cov.mean(axis=0))
    R = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    T = np.eye(4)
    T[:3, :3] = R
    C0 = np.diag([1.0, 2.0, 3.0]).reshape(9)
    out = orc.transform_cov(T, C0[None, :])
    assert out.shape[0] == 1 and np.allclose(out[0].reshape(3, 3), np.diag([2.0, 1.0, 3.0]))        # R C R^T swaps the x / y variances
