"""Bit-exact parity of the open3d_slam-side operators (include/o3s_cloud_ops.h) against the oracle.  MI355X only."""
import numpy as np
import pytest

from oracle import oracle as orc
from open3d_slam_advanced_rss_2024_public_amd import cloud_ops as ops

pytestmark = pytest.mark.gpu


def test_voxel_idx_and_hash_bit_exact():
    rng = np.random.default_rng(0)
    p = rng.uniform(-500, 500, (200_000, 3))
    k = rng.integers(-3000, 3000, (50_000, 3))
    p[:50_000] = k * 0.1          # points exactly on nominal cell boundaries
    p[50_000:50_010] = [[-0.05, 0.3, 0.0]] * 10
    for v in (0.1, 0.25, 0.02, 1.0 / 3.0):
        a = ops.getVoxelIdx(p, v)
        b = orc.voxel_idx(p, v)
        assert np.array_equal(a, b)
    idx = ops.getVoxelIdx(p, 0.1)
    assert np.array_equal(ops.voxelHash(idx), orc.voxel_hash(idx))
    assert ops.getVoxelIdx(np.array([[-0.05, 0.3, 0.0]]), 0.1).tolist() == [[-1, int(np.floor(0.3 * (1.0 / 0.1))), 0]]


@pytest.mark.parametrize("kind,args", [("MaxRadius", (15.0,)), ("MinRadius", (5.0,)), ("MinMaxRadius", (5.0, 15.0)),
                                       ("Cylinder", (10.0, -3.0, 4.0)), ("CroppingVolume", ())])
@pytest.mark.parametrize("invert", [False, True])
def test_crop_order_preserving_bit_exact(kind, args, invert):
    rng = np.random.default_rng(1)
    p = rng.uniform(-20, 20, (100_003, 3))
    n = rng.normal(size=p.shape)
    c = (1.0, -2.0, 0.5)
    a = list(args) + [0.0] * (3 - len(args))
    g = ops.croppingVolumeFactory(kind, *a, centre=c, invert=invert)
    o = orc.make_cropper("Base" if kind == "CroppingVolume" else kind, *a, centre=c, invert=invert)
    gp, gn = ops.crop(g, p, n)
    m = orc.crop_mask(o, p)
    assert np.array_equal(gp, p[m]) and np.array_equal(gn, n[m])
    gp2, gn2 = ops.crop(g, p, None)
    assert np.array_equal(gp2, p[m]) and gn2 is None


def _as_dict(pts, nrm, idx):
    return {tuple(k): (pp, None if nrm is None else nn) for k, pp, nn in zip(idx, pts, nrm if nrm is not None else pts)}


def test_voxelize_within_crop_bit_exact():
    rng = np.random.default_rng(2)
    p = rng.uniform(-6, 6, (120_000, 3))
    n = rng.normal(size=p.shape)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    n[5] = np.nan
    n[77, 1] = np.nan
    for voxel in (0.25, 0.1):
        g = ops.croppingVolumeFactory("MaxRadius", 4.0, centre=(0.5, 0.0, -0.5))
        o = orc.make_cropper("MaxRadius", 4.0, centre=(0.5, 0.0, -0.5))
        gp, gn, gi = ops.voxelizeWithinCroppingVolume(voxel, g, p, n)
        op, on, oi = orc.voxelize_within_crop(o, voxel, p, n)
        assert len(gp) == len(op)
        npass = int((oi[:, 0] == np.iinfo(np.int32).min).sum())
        # pass-through block: identical, in input order
        assert np.array_equal(gp[:npass], op[:npass]) and np.array_equal(gn[:npass], on[:npass], equal_nan=True)
        assert np.all(gi[:npass] == np.iinfo(np.int32).min)
        # voxel block: same set, bit-identical means (per-voxel sums run in input order on both sides)
        go = np.lexsort((gi[npass:, 0], gi[npass:, 1], gi[npass:, 2]))
        oo = np.lexsort((oi[npass:, 0], oi[npass:, 1], oi[npass:, 2]))
        assert np.array_equal(gi[npass:][go], oi[npass:][oo])
        assert np.array_equal(gp[npass:][go], op[npass:][oo])
        assert np.array_equal(gn[npass:][go], on[npass:][oo], equal_nan=True)
        # the library's own order is ascending (z, y, x)
        assert np.array_equal(go, np.arange(len(go)))
    # voxel_size <= 0 returns the cloud unchanged (helpers.cpp:122-125); no normals path
    g = ops.croppingVolumeFactory("MaxRadius", 4.0)
    gp, gn, gi = ops.voxelizeWithinCroppingVolume(0.0, g, p[:1000], None)
    assert np.array_equal(gp, p[:1000]) and gn is None
    gp, gn, gi = ops.voxelizeWithinCroppingVolume(0.5, g, p[:5000], None)
    op, on, oi = orc.voxelize_within_crop(orc.make_cropper("MaxRadius", 4.0), 0.5, p[:5000], None)
    assert len(gp) == len(op)


def test_o3d_voxel_downsample_and_conversion_bit_exact():
    rng = np.random.default_rng(3)
    p = rng.uniform(-30, 30, (130_000, 3))
    n = rng.normal(size=p.shape)
    gp, gn, gi = ops.voxelize(0.25, p, n)
    op, on, oi = orc.voxel_downsample_o3d(0.25, p, n)
    assert len(gp) == len(op)
    go = np.lexsort((gi[:, 0], gi[:, 1], gi[:, 2]))
    oo = np.lexsort((oi[:, 0], oi[:, 1], oi[:, 2]))
    assert np.array_equal(gi[go], oi[oo]) and np.array_equal(gp[go], op[oo]) and np.array_equal(gn[go], on[oo])
    xyzw, nn = ops.open3dToPointmatcher(p, n)
    assert np.array_equal(xyzw[:, :3], p.astype(np.float32)) and np.all(xyzw[:, 3] == 1) and np.array_equal(nn, n.astype(np.float32))
    oxyzw, onn = orc.o3d_to_pm(p, n)
    assert np.array_equal(xyzw, oxyzw) and np.array_equal(nn, onn)


def test_empty_and_tiny_inputs():
    g = ops.croppingVolumeFactory("MaxRadius", 1.0)
    e = np.zeros((0, 3))
    assert ops.getVoxelIdx(e, 0.1).shape == (0, 3)
    assert ops.crop(g, e)[0].shape == (0, 3)
    assert ops.voxelizeWithinCroppingVolume(0.1, g, e)[0].shape == (0, 3)
    assert ops.voxelize(0.1, e)[0].shape == (0, 3)
    one = np.array([[0.2, 0.2, 0.2]])
    gp, _, gi = ops.voxelizeWithinCroppingVolume(0.5, g, one)
    assert np.array_equal(gp, one) and gi.tolist() == [[0, 0, 0]]
    gp, _, gi = ops.voxelizeWithinCroppingVolume(0.5, g, one + 5.0)   # outside the cropper -> pass-through
    assert np.array_equal(gp, one + 5.0) and gi[0, 0] == np.iinfo(np.int32).min


from open3d_slam_advanced_rss_2024_public_amd import cloud_ops as co, synthetic as syn


def _coloured_cloud(n=30000, seed=3):
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-6, 6, (n, 3))
    pts[::97] = np.round(pts[::97] * 4) / 4           # points on voxel boundaries
    nrm = rng.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    col = rng.uniform(0, 1, (n, 3))
    col[::13] = rng.uniform(-2, 3, (len(col[::13]), 3))   # "invalid" colours: the reference's isValidColor accepts them all the same
    A = rng.normal(size=(n, 3, 3))
    cov = np.einsum("nij,nkj->nik", A, A).reshape(n, 9)    # symmetric positive semi-definite
    return pts, nrm, col, cov


def test_colours_and_covariances_ride_along_bit_exact():
    """helpers.cpp:141-187 / croppers.cpp:76-106 with the colour and covariance lanes: crop copies them, voxelise-within-crop
    keeps the LAST colour of a voxel in input order and the mean covariance, Open3D's VoxelDownSample averages both."""
    pts, nrm, col, cov = _coloured_cloud()
    c = co.croppingVolumeFactory("MaxRadius", 4.5, centre=(0.5, -0.25, 0.0))
    oc = orc.make_cropper("MaxRadius", 4.5, centre=(0.5, -0.25, 0.0))
    m = orc.crop_mask(oc, pts)
    p, n_, cl, cv = co.crop_attr(c, pts, nrm, col, cov)
    assert np.array_equal(p, pts[m]) and np.array_equal(n_, nrm[m]) and np.array_equal(cl, col[m]) and np.array_equal(cv, cov[m])
    # voxelise within the crop: compare as sets keyed by voxel index (pass-through part in input order)
    gp, gn, gcol, gcov, gidx = co.voxelizeWithinCroppingVolume_attr(0.25, c, pts, nrm, col, cov)
    op, on, oidx = orc.voxelize_within_crop(oc, 0.25, pts, nrm)
    ocol, ocov = orc.voxelize_attrs(0, oc, 0.25, pts, col, cov)
    k = int((oidx[:, 0] == np.iinfo(np.int32).min).sum())
    order = np.concatenate([np.arange(k), np.lexsort((oidx[k:, 0], oidx[k:, 1], oidx[k:, 2])) + k])
    assert np.array_equal(gidx, oidx[order]) and np.array_equal(gp, op[order]) and np.array_equal(gn, on[order])
    assert np.array_equal(gcol, ocol[order]) and np.array_equal(gcov, ocov[order])
    # the wrappers without attributes are unchanged
    p0, n0, i0 = co.voxelizeWithinCroppingVolume(0.25, c, pts, nrm)
    assert np.array_equal(p0, gp) and np.array_equal(n0, gn) and np.array_equal(i0, gidx)
    # Open3D voxel down-sample: mean colour, mean covariance
    gp, gn, gcol, gcov, gidx = co.voxelize_attr(0.3, pts, nrm, col, cov)
    op, on, oidx = orc.voxel_downsample_o3d(0.3, pts, nrm)
    ocol, ocov = orc.voxelize_attrs(1, None, 0.3, pts, col, cov)
    order = np.lexsort((oidx[:, 0], oidx[:, 1], oidx[:, 2]))
    assert np.array_equal(gidx, oidx[order]) and np.array_equal(gp, op[order]) and np.array_equal(gcol, ocol[order]) and np.array_equal(gcov, ocov[order])


def test_transform_with_covariances_and_identity_doubling():
    """o3d_slam::transform (helpers.cpp:283-318): R C R^T, and the doubled output for an almost-identity pose."""
    pts, nrm, col, cov = _coloured_cloud(5000, seed=4)
    T = syn.make_T(syn.rot_axis_angle([0.3, -0.5, 0.8], 0.7), np.array([3.0, -2.0, 0.5]))
    gp, gn, gc = co.transform(T, pts, nrm, cov)
    op, on = orc.transform_cloud(T, pts, nrm)
    oc = orc.transform_cov(T, cov)
    assert np.array_equal(gp, op) and np.array_equal(gn, on) and np.array_equal(gc, oc) and len(gp) == len(pts)
    Ti = np.eye(4)
    Ti[0, 3] = 5e-5
    gp, gn, gc = co.transform(Ti, pts, nrm, cov)
    op, on = orc.transform_cloud(Ti, pts, nrm)
    oc = orc.transform_cov(Ti, cov)
    assert len(gp) == 2 * len(pts) and np.array_equal(gp, op) and np.array_equal(gn, on) and np.array_equal(gc, oc)


@pytest.mark.parametrize("n", [1, 7, 2047, 2048, 2049, 5000, 13337, 16384, 16385, 20481, 40000, 53211, 65536, 131071, 131072, 131073, 200000, 262143, 262144, 300000])
def test_pair_sort_of_the_work_areas_is_a_stable_sort(hooks_lib, n):
    """csrc/cloud_dev.h sort_pairs (rocPRIM's merge path below 262 144 pairs, Onesweep above; the library's own small sort was measured
    slower in round 5 and removed): the stable order of the keys, with many ties and a key of all ones among them."""
    import ctypes as C
    rng = np.random.default_rng(n)
    keys = rng.integers(0, max(2, n // 3), n).astype(np.uint64) << np.uint64(17)
    keys[rng.integers(0, n, max(1, n // 50))] = np.uint64(0xFFFFFFFFFFFFFFFF)
    vals = rng.permutation(n).astype(np.uint32)
    ko, vo = np.empty_like(keys), np.empty_like(vals)
    f = hooks_lib.o3s_test_sort_pairs
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    assert f(0, keys.ctypes.data, vals.ctypes.data, n, 64, ko.ctypes.data, vo.ctypes.data) == 0
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(ko, keys[order]) and np.array_equal(vo, vals[order])
