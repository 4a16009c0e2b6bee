"""Oracle pins for the dense map (o3d_slam::VoxelizedPointCloud) and its space carving.  The reference holds no test for
these functions, so the expectations below are worked out by hand from the reference's source
(open3d_slam/src/Voxel.cpp, VoxelHashMap.cpp:13-46, helpers.cpp:360-390, Submap.cpp:146-157)."""
import itertools

import numpy as np

from oracle import oracle as orc


def test_insert_and_to_point_cloud():
    """Voxel.cpp:66-114: key = floor(p * (1/voxel)); position / normal sums in insertion order; mean = sum / count
    (the normal is NOT re-normalised)."""
    m = orc.DenseMap(0.5)
    assert m.size() == 0
    p = np.array([[0.1, 0.1, 0.1], [0.2, 0.3, 0.4], [-0.1, 0.0, 0.0], [0.4, 0.45, 0.49]])
    n = np.array([[0.0, 0.0, 1.0], [0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    m.insert(p, n)
    assert m.size() == 2
    pts, nrm, keys, cnt = m.to_point_cloud()
    assert keys.tolist() == [[-1, 0, 0], [0, 0, 0]] and cnt.tolist() == [1, 3]
    exp0 = [((0.1 + 0.2) + 0.4) / 3.0, ((0.1 + 0.3) + 0.45) / 3.0, ((0.1 + 0.4) + 0.49) / 3.0]
    assert pts[1].tolist() == exp0 and pts[0].tolist() == [-0.1, 0.0, 0.0]
    assert nrm[1].tolist() == [0.0, 1.0 / 3.0, 2.0 / 3.0]
    # a second insert keeps adding to the same sums
    m.insert(np.array([[0.3, 0.3, 0.3]]), np.array([[0.0, 0.0, 1.0]]))
    pts, nrm, keys, cnt = m.to_point_cloud()
    assert cnt.tolist() == [1, 4] and pts[1, 0] == (((0.1 + 0.2) + 0.4) + 0.3) / 4.0


def test_insert_without_normals_then_with():
    m = orc.DenseMap(1.0)
    m.insert(np.array([[0.5, 0.5, 0.5]]))
    assert not m.has_normals and m.to_point_cloud()[1] is None
    m.insert(np.array([[0.6, 0.5, 0.5]]), np.array([[0.0, 0.0, 2.0]]))
    assert m.has_normals
    pts, nrm, keys, cnt = m.to_point_cloud()
    assert cnt.tolist() == [2] and nrm[0].tolist() == [0.0, 0.0, 1.0]  # normal sum / point count (Voxel.cpp:21-23)


def test_transform_maps_sums_as_points_and_keeps_keys():
    """Voxel.cpp:49-64: both sums go through Transform * Vector3d (rotation AND translation); keys stay."""
    m = orc.DenseMap(1.0)
    m.insert(np.array([[0.5, 0.5, 0.5], [0.25, 0.5, 0.5]]), np.array([[0.0, 0.0, 1.0], [0.0, 0.0, 1.0]]))
    T = np.eye(4)
    T[:3, :3] = [[0, -1, 0], [1, 0, 0], [0, 0, 1]]  # +90 deg about z
    T[:3, 3] = [10.0, 20.0, 30.0]
    m.transform(T)
    pts, nrm, keys, cnt = m.to_point_cloud()
    assert keys.tolist() == [[0, 0, 0]] and cnt.tolist() == [2]
    # sum p = (0.75, 1, 1) -> R s + t = (-1 + 10, 0.75 + 20, 1 + 30); mean = that / 2
    assert pts[0].tolist() == [9.0 / 2, 20.75 / 2, 31.0 / 2]
    assert nrm[0].tolist() == [10.0 / 2, 20.0 / 2, 32.0 / 2]


def test_remove_duplicate_points():
    """Voxel.cpp:162-192: the first point of every voxel survives, order kept."""
    p = np.array([[0.01, 0, 0], [0.02, 0, 0], [0.11, 0, 0], [0.03, 0, 0], [-0.01, 0, 0], [0.19, 0.0, 0.0]])
    assert orc.remove_duplicate_points(p, 0.1).tolist() == [True, False, True, False, True, False]


def test_neighbourhood_keys():
    """VoxelHashMap.cpp:13-46."""
    # radius <= 0: the centre key only (dividing form of getVoxelIdx)
    assert orc.voxels_within_neighborhood([0.26, -0.01, 0.0], 0.0, 0.1).tolist() == [[2, -1, 0]]
    # radius == voxel: offsets {-0.1, 0, 0.1} per axis; every test point sits 0.01 from its voxel centre on each axis,
    # so all 27 are within the radius, emitted x-major / z-fastest; the centre is among them (no extra entry)
    k = orc.voxels_within_neighborhood([0.26, 0.26, 0.26], 0.1, 0.1)
    assert k.tolist() == [list(t) for t in itertools.product([1, 2, 3], repeat=3)]
    # radius = voxel / 2: offsets {-0.05, 0.05}; each test point is 0.03 off its centre per axis, |.| = 0.052 > 0.05:
    # nothing qualifies and the centre key is appended
    assert orc.voxels_within_neighborhood([0.27, 0.27, 0.27], 0.05, 0.1).tolist() == [[2, 2, 2]]
    # radius = 2 voxels: 5 offsets per axis, all 125 test points lie inside their own voxel (<= 0.044 from its centre)
    k = orc.voxels_within_neighborhood([0.26, 0.26, 0.26], 0.1, 0.05)
    assert k.tolist() == [list(t) for t in itertools.product([3, 4, 5, 6, 7], repeat=3)]


def _row_map(voxel, ks, js):
    m = orc.DenseMap(voxel)
    pts = np.array([[(k + 0.5) * voxel, (j + 0.5) * voxel, 0.5 * voxel] for k in ks for j in js])
    m.insert(pts)
    return m


def test_carve_centre_only():
    """helpers.cpp:360-390 with radius = voxel / 2 (only the centre voxel of each stop, see above): a ray along +x from
    (0.03, 0.03, 0.03) to x = 3.03 stops every 0.1 m while distance < max(0.1, min(3 - 0.1, 20)) = 2.9, i.e. at
    x = 0.03 + 0.1 n for n = 0..28 -> voxels (n, 0, 0), n = 0..28, disappear."""
    m = _row_map(0.1, range(50), [0])
    m.insert(np.array([[2.05, 3.05, 0.05]]))
    scan = np.array([[3.03, 0.03, 0.03], [3.035, 0.03, 0.03]])  # the second point shares the first one's voxel: dropped
    assert orc.remove_duplicate_points(scan, 0.1).tolist() == [True, False]
    removed = m.carve(scan, [0.03, 0.03, 0.03], neighborhood_radius=0.05, max_length=20.0, truncation=0.1)
    assert removed == 29 and m.size() == 51 - 29
    keys = m.to_point_cloud()[2]
    assert sorted(keys[:, 0].tolist()) == sorted(list(range(29, 50)) + [20]) and [20, 30, 0] in keys.tolist()


def test_carve_full_neighbourhood_and_limits():
    """radius = voxel = 0.1: stops every 0.2 m at x = 0.05 + 0.2 n, n = 0..14 (distance < 2.9); each stop clears keys
    (2n - 1 .. 2n + 1) x (-1..1) x (-1..1).  Rows j = -1, 0, 1 lose k = 0..29, row j = 2 is untouched."""
    m = _row_map(0.1, range(50), [-1, 0, 1, 2])
    scan = np.array([[3.05, 0.05, 0.05]])
    removed = m.carve(scan, [0.05, 0.05, 0.05], neighborhood_radius=0.1, max_length=20.0, truncation=0.1)
    assert removed == 90 and m.size() == 200 - 90
    # maxRaytracingLength caps the march: distance < max(0.2, min(2.9, 1.0)) = 1.0 -> n = 0..4 -> k = 0..9
    m = _row_map(0.1, range(50), [0])
    assert m.carve(scan, [0.05, 0.05, 0.05], neighborhood_radius=0.1, max_length=1.0, truncation=0.1) == 10
    # a return closer than the truncation distance still marches one step (the max with the step size)
    m = _row_map(0.1, range(50), [0])
    assert m.carve(np.array([[0.1, 0.05, 0.05]]), [0.05, 0.05, 0.05], neighborhood_radius=0.1, max_length=20.0, truncation=0.1) == 2
    # an empty map or an empty scan is a no-op
    assert orc.DenseMap(0.1).carve(scan, [0, 0, 0]) == 0 and m.carve(np.zeros((0, 3)), [0, 0, 0]) == 0
