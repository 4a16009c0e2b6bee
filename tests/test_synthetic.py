"""The seeded synthetic inputs of SURVEY.md 8(d): sanity of the generators the parity tests and the benches rely on."""
import numpy as np

from open3d_slam_advanced_rss_2024_public_amd import synthetic as syn


def test_lidar_ray_cast_returns_lie_on_the_world():
    """64 x 2048 spherical grid (config 5): ~130 k returns, each on the rectangle it reports the normal of, none beyond
    max_range, deterministic for a seed."""
    world = syn.make_world(20000.0, seed=11)
    T = syn.make_T(syn.rot_axis_angle([0, 0, 1], 0.4), np.array([3.0, -2.0, 1.5]))
    p, n = syn.make_lidar_scan(world, T, sigma=0.0, seed=1)
    assert 100000 < p.shape[0] <= 64 * 2048 and p.dtype == np.float32
    r = np.linalg.norm(p.astype(np.float64), axis=1)
    assert r.max() <= 60.0 + 1e-3 and r.min() > 0.05
    pw = p.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    nw = n.astype(np.float64) @ T[:3, :3].T
    # every return lies on an axis-aligned rectangle of the world: its normal is a unit axis vector and the point's
    # coordinate along that axis is one of the planes' offsets
    axis = np.argmax(np.abs(nw), axis=1)
    assert np.allclose(np.abs(nw[np.arange(len(nw)), axis]), 1.0, atol=1e-6)
    offsets = {a: np.unique(np.round(world.centres[np.abs(world.normals[:, a]) > 0.5, a], 6)) for a in range(3)}
    for a in range(3):
        c = pw[axis == a, a]
        assert c.size > 0
        assert np.abs(c[:, None] - offsets[a][None, :]).min(axis=1).max() < 1e-3
    # the rays see the nearest surface: nothing is returned from behind the floor or the ceiling
    assert pw[:, 2].min() > -1e-3 and pw[:, 2].max() < world.size[2] + 1e-3
    p2, _ = syn.make_lidar_scan(world, T, sigma=0.0, seed=1)
    assert np.array_equal(p, p2)
    pn, _ = syn.make_lidar_scan(world, T, sigma=0.01, seed=2)
    assert pn.shape == p.shape and 0.005 < np.std(np.linalg.norm(pn.astype(np.float64), axis=1) - r) < 0.02
