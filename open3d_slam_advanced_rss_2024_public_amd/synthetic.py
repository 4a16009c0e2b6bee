"""Seeded synthetic fixtures for the scan-to-map ICP path (numpy only; no reference code is run).

Two families:

* ``box_cloud`` / ``conditioning_cases`` restate the reference's registration test fixtures
  (libpointmatcher/pointmatcher/PointCloudGenerator.cpp:283-375 ``generateUniformlySampledPlane/Box``,
  libpointmatcher/pointmatcher/testing/RegistrationTestCase.cpp:7-64 and the 20 pose cases of
  libpointmatcher/utest/ui/icp/Conditioning.cpp:31-255) with a fixed seed — the reference seeds them from the
  clock, so only the tolerance contract is reproducible, not the vectors.
* ``make_world`` / ``make_scan_pair`` build the benchmark inputs of SURVEY.md §8(d): an axis-aligned
  "room + pillars" world with analytic normals, a voxel-snapped map of exactly M points, a noisy scan of N points in
  the sensor frame, a ground-truth pose and a perturbed initial guess.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np


# ------------------------------------------------------------------------------------------------
# small SE(3) helpers (float64; callers cast)
# ------------------------------------------------------------------------------------------------
def rot_axis_angle(axis, angle) -> np.ndarray:
    a = np.asarray(axis, np.float64)
    n = np.linalg.norm(a)
    if n == 0 or angle == 0:
        return np.eye(3)
    a = a / n
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + math.sin(angle) * K + (1 - math.cos(angle)) * (K @ K)


def rot_rpy_deg(roll, pitch, yaw) -> np.ndarray:
    """buildQuaternionFromRPY (libpointmatcher/pointmatcher/testing/utils_geometry.cpp:19-28): Rx * Ry * Rz."""
    return (rot_axis_angle([1, 0, 0], math.radians(roll)) @ rot_axis_angle([0, 1, 0], math.radians(pitch))
            @ rot_axis_angle([0, 0, 1], math.radians(yaw)))


def make_T(R=None, t=None) -> np.ndarray:
    T = np.eye(4)
    if R is not None:
        T[:3, :3] = R
    if t is not None:
        T[:3, 3] = t
    return T


def transform_cloud(T, xyz, normals=None):
    """fp32 rigid transform, same operation order as RigidTransformation::inPlaceCompute (sum over k = 0..3)."""
    T = np.asarray(T, np.float32)
    p = np.asarray(xyz, np.float32)
    out = np.empty_like(p)
    for r in range(3):
        s = T[r, 0] * p[:, 0]
        s = s + T[r, 1] * p[:, 1]
        s = s + T[r, 2] * p[:, 2]
        s = s + T[r, 3] * np.float32(1.0)
        out[:, r] = s
    nout = None
    if normals is not None:
        n = np.asarray(normals, np.float32)
        nout = np.empty_like(n)
        for r in range(3):
            s = T[r, 0] * n[:, 0]
            s = s + T[r, 1] * n[:, 1]
            s = s + T[r, 2] * n[:, 2]
            nout[:, r] = s
    return out, nout


# ------------------------------------------------------------------------------------------------
# reference-style box fixtures
# ------------------------------------------------------------------------------------------------
def _plane(rng, dims, n, centre, flip):
    """generateUniformlySampledPlane: uniform in [-dims, dims] per axis (a zero dim pins that axis), analytic normal."""
    pts = np.zeros((n, 3), np.float32)
    for a in range(3):
        if dims[a] != 0:
            pts[:, a] = rng.uniform(-dims[a], dims[a], n).astype(np.float32)
    normal = np.array([1.0 if d == 0 else 0.0 for d in dims], np.float32)
    if flip:
        normal = -normal
    nrm = np.tile(normal, (n, 1))
    pts = pts + np.asarray(centre, np.float32)
    return pts, nrm


def box_cloud(length, width, height, n_points, seed, T=None):
    """generateUniformlySampledBox (PointCloudGenerator.cpp:329-375): 6 faces, n/6 points each, remainder on -Z."""
    rng = np.random.default_rng(seed)
    per = n_points // 6
    L, W, H = np.float32(length), np.float32(width), np.float32(height)
    faces = [
        ((0, W * 0.5, H * 0.5), (L * 0.5, 0, 0), False),
        ((0, W * 0.5, H * 0.5), (-L * 0.5, 0, 0), True),
        ((L * 0.5, 0, H * 0.5), (0, W * 0.5, 0), False),
        ((L * 0.5, 0, H * 0.5), (0, -W * 0.5, 0), True),
        ((L * 0.5, W * 0.5, 0), (0, 0, H * 0.5), False),
        ((L * 0.5, W * 0.5, 0), (0, 0, -H * 0.5), True),
    ]
    pts, nrm = [], []
    for k, (dims, centre, flip) in enumerate(faces):
        n = per if k < 5 else n_points - 5 * per
        p, q = _plane(rng, dims, n, centre, flip)
        pts.append(p)
        nrm.append(q)
    pts = np.concatenate(pts)
    nrm = np.concatenate(nrm)
    if T is not None:
        pts, nrm = transform_cloud(T, pts, nrm)
    return pts, nrm


@dataclass
class RegistrationCase:
    name: str
    T_origin_ref: np.ndarray
    T_ref_read: np.ndarray
    T_origin_read: np.ndarray
    initial_guess: np.ndarray
    ref_xyz: np.ndarray
    ref_normals: np.ndarray
    read_xyz: np.ndarray
    read_normals: np.ndarray


# (name, origin->ref translation, origin->ref rpy, ref->read translation, ref->read rpy): Conditioning.cpp:54-253
_CASE_TABLE = [
    ("BoxClouds_NoReferenceShift_NoDisplacement", (0, 0, 0), (0, 0, 0), (0, 0, 0), (0, 0, 0)),
    ("BoxClouds_1mXReferenceShift_NoDisplacement", (1, 0, 0), (0, 0, 0), (0, 0, 0), (0, 0, 0)),
    ("BoxClouds_1000mXReferenceShift_NoDisplacement", (1000, 0, 0), (0, 0, 0), (0, 0, 0), (0, 0, 0)),
    ("BoxClouds_10DegYawReferenceShift_NoDisplacement", (0, 0, 0), (0, 0, 10), (0, 0, 0), (0, 0, 0)),
    ("BoxClouds_30DegYawReferenceShift_NoDisplacement", (0, 0, 0), (0, 0, 30), (0, 0, 0), (0, 0, 0)),
    ("BoxClouds_1mX10DegYawReferenceShift_NoDisplacement", (1, 0, 0), (0, 0, 10), (0, 0, 0), (0, 0, 0)),
    ("BoxClouds_1mX30DegYawReferenceShift_NoDisplacement", (1, 0, 0), (0, 0, 30), (0, 0, 0), (0, 0, 0)),
    ("BoxClouds_NoReferenceShift_10DegYawDisplacement", (0, 0, 0), (0, 0, 0), (0, 0, 0), (0, 0, 10)),
    ("BoxClouds_NoYawReferenceShift_30DegYawDisplacement", (0, 0, 0), (0, 0, 0), (0, 0, 0), (0, 0, 30)),
    ("BoxClouds_1mXReferenceShift_10DegYawDisplacement", (1, 0, 0), (0, 0, 0), (0, 0, 0), (0, 0, 10)),
    ("BoxClouds_100mX-3000mZReferenceShift_10DegYawDisplacement", (100, 0, -3000), (0, 0, 0), (0, 0, 0), (0, 0, 10)),
    ("BoxClouds_100mX+3000mY+30DegYawReferenceShift_10DegYawDisplacement", (100, 3000, 0), (0, 0, 0), (0, 0, 0), (0, 0, 30)),
    ("BoxClouds_10000mX+3000mY-3500Z+30DegPitchRollYawYawReferenceShift_10DegYawDisplacement", (10000, 3000, -3500),
     (0, 0, 0), (0, 0, 0), (30, 30, 30)),
    ("BoxClouds_30DegYawReferenceShift_10DegYawDisplacement", (0, 0, 0), (0, 0, 30), (0, 0, 0), (0, 0, 10)),
    ("BoxClouds_1mX30DegYawReferenceShift_10DegYawDisplacement", (1, 0, 0), (0, 0, 30), (0, 0, 0), (0, 0, 10)),
    ("BoxClouds_2mX30DegYawReferenceShift_10DegYawDisplacement", (2, 0, 0), (0, 0, 30), (0, 0, 0), (0, 0, 10)),
    ("BoxClouds_2mX30DegYawReferenceShift_-1mY10DegYawDisplacement", (2, 0, 0), (0, 0, 30), (0, -1, 0), (0, 0, 10)),
    ("BoxClouds_10mX30DegYawReferenceShift_-1mY10DegYawDisplacement", (10, 0, 0), (0, 0, 30), (0, -1, 0), (0, 0, 10)),
    ("BoxClouds_10mY-30DegYawReferenceShift_-1mY10DegYawDisplacement", (0, 10, 0), (0, 0, -30), (0, -1, 0), (0, 0, 10)),
    ("BoxClouds_10mY-30DegYawReferenceShift_-1mY-50DegYawDisplacement", (0, 10, 0), (0, 0, -30), (0, -1, 0), (0, 0, -50)),
]


def conditioning_cases(n_points=10000, scale=1.0, trans_noise_std=0.0, rot_noise_std_deg=0.0, same_clouds=True,
                       seed=20240807):
    """setUpTestCases (Conditioning.cpp:31-255) + RegistrationTestCase ctor, seeded."""
    rng = np.random.default_rng(seed)
    # buildRandomVectorFromStdDev: Vector::Random() (uniform [-1,1]) * std
    t_err = rng.uniform(-1, 1, 3) * trans_noise_std
    ang = rng.normal(0, math.radians(rot_noise_std_deg)) if rot_noise_std_deg > 0 else 0.0
    axis = rng.uniform(-1, 1, 3)
    T_err = make_T(rot_axis_angle(axis, ang), t_err)
    cases = []
    for k, (name, t_or, rpy_or, t_rr, rpy_rr) in enumerate(_CASE_TABLE):
        T_origin_ref = make_T(rot_rpy_deg(*rpy_or), np.asarray(t_or, np.float64) * scale)
        T_ref_read = make_T(rot_rpy_deg(*rpy_rr), np.asarray(t_rr, np.float64) * scale)
        T_origin_read = T_origin_ref @ T_ref_read
        guess = T_origin_ref @ T_err
        L, W, H = 1.0 * scale, 3.0 * scale, 5.0 * scale
        T_read = np.linalg.inv(T_ref_read)
        if same_clouds:
            p, n = box_cloud(L, W, H, n_points, seed + 1000 + k)
            ref_xyz, ref_n = transform_cloud(T_origin_ref, p, n)
            read_xyz, read_n = transform_cloud(T_read, p, n)
        else:
            ref_xyz, ref_n = box_cloud(L, W, H, n_points, seed + 2000 + k, T_origin_ref)
            read_xyz, read_n = box_cloud(L, W, H, n_points, seed + 3000 + k, T_read)
        cases.append(RegistrationCase(name, T_origin_ref, T_ref_read, T_origin_read, guess, ref_xyz, ref_n, read_xyz, read_n))
    return cases


# ------------------------------------------------------------------------------------------------
# benchmark world
# ------------------------------------------------------------------------------------------------
@dataclass
class World:
    centres: np.ndarray  # (F,3)
    u: np.ndarray        # (F,3) half-extent vector 1
    v: np.ndarray        # (F,3) half-extent vector 2
    normals: np.ndarray  # (F,3)
    areas: np.ndarray    # (F,)
    size: tuple          # (L, W, H)


def make_world(target_area: float, height: float = 6.0, pitch: float = 12.0, seed: int = 1234) -> World:
    """Room (floor, ceiling, 4 walls) + a jittered grid of box pillars (one per `pitch` x `pitch` cell, so every
    15 m scan disc sees vertical structure in both horizontal directions); total surface area ~ target_area."""
    rng = np.random.default_rng(seed)
    H = height
    pillar_w = 3.0
    # area(L) = 2 L^2 + 4 L H + (L/pitch)^2 * 4 * pillar_w * H
    k = 4.0 * pillar_w * H / (pitch * pitch)
    a = 2.0 + k
    b = 4.0 * H
    c = -target_area
    L = (-b + math.sqrt(b * b - 4 * a * c)) / (2 * a)
    g = max(int(round(L / pitch)), 1)
    if L < 2 * pitch:
        pillar_w = L / 8.0
    W = L
    cs, us, vs, ns = [], [], [], []

    def add(c_, u_, v_, n_):
        cs.append(c_)
        us.append(u_)
        vs.append(v_)
        ns.append(n_)

    add((0, 0, 0), (L / 2, 0, 0), (0, W / 2, 0), (0, 0, 1))        # floor, normal up
    add((0, 0, H), (L / 2, 0, 0), (0, W / 2, 0), (0, 0, -1))       # ceiling, normal down
    add((L / 2, 0, H / 2), (0, W / 2, 0), (0, 0, H / 2), (-1, 0, 0))
    add((-L / 2, 0, H / 2), (0, W / 2, 0), (0, 0, H / 2), (1, 0, 0))
    add((0, W / 2, H / 2), (L / 2, 0, 0), (0, 0, H / 2), (0, -1, 0))
    add((0, -W / 2, H / 2), (L / 2, 0, 0), (0, 0, H / 2), (0, 1, 0))
    cell = L / g
    hw = pillar_w / 2
    jit = max(cell / 2 - pillar_w, 0.0) * 0.5
    for i in range(g):
        for j in range(g):
            cx = -L / 2 + (i + 0.5) * cell + rng.uniform(-jit, jit)
            cy = -W / 2 + (j + 0.5) * cell + rng.uniform(-jit, jit)
            add((cx + hw, cy, H / 2), (0, hw, 0), (0, 0, H / 2), (1, 0, 0))
            add((cx - hw, cy, H / 2), (0, hw, 0), (0, 0, H / 2), (-1, 0, 0))
            add((cx, cy + hw, H / 2), (hw, 0, 0), (0, 0, H / 2), (0, 1, 0))
            add((cx, cy - hw, H / 2), (hw, 0, 0), (0, 0, H / 2), (0, -1, 0))
    cs, us, vs, ns = (np.asarray(x, np.float64) for x in (cs, us, vs, ns))
    areas = 4.0 * np.linalg.norm(us, axis=1) * np.linalg.norm(vs, axis=1)
    return World(cs, us, vs, ns, areas, (L, W, H))


def sample_world(world: World, n: int, rng) -> tuple:
    f = rng.choice(len(world.areas), size=n, p=world.areas / world.areas.sum())
    a = rng.uniform(-1, 1, n)[:, None]
    b = rng.uniform(-1, 1, n)[:, None]
    pts = world.centres[f] + a * world.u[f] + b * world.v[f]
    return pts, world.normals[f]


def voxel_keys(pts: np.ndarray, voxel: float) -> np.ndarray:
    """getVoxelIdx with the reciprocal form (VoxelHashMap.hpp:43-51), fp64."""
    inv = 1.0 / voxel
    return np.floor(pts * inv).astype(np.int64)


def make_map(world: World, M: int, voxel: float, seed: int = 1234, oversample: float = 3.0):
    """Exactly M map points: area-uniform samples voxel-averaged on the absolute grid, seeded subset of M voxels."""
    rng = np.random.default_rng(seed)
    n_vox = world.areas.sum() / (voxel * voxel)
    n_s = int(max(oversample * n_vox, 2 * M))
    pts, nrm = sample_world(world, n_s, rng)
    k = voxel_keys(pts, voxel)
    # one map point per (voxel, face-normal): pack key
    off = k.min(axis=0)
    kk = k - off
    dims = kk.max(axis=0) + 1
    nid = (np.argmax(np.abs(nrm), axis=1) * 2 + (nrm.sum(axis=1) > 0)).astype(np.int64)
    lin = ((kk[:, 0] * dims[1] + kk[:, 1]) * dims[2] + kk[:, 2]) * 6 + nid
    order = np.argsort(lin, kind="stable")
    lin_s = lin[order]
    first = np.concatenate([[True], lin_s[1:] != lin_s[:-1]])
    starts = np.flatnonzero(first)
    counts = np.diff(np.concatenate([starts, [len(lin_s)]]))
    mean_p = np.add.reduceat(pts[order], starts, axis=0) / counts[:, None]
    mean_n = nrm[order][starts]
    if len(starts) < M:
        raise ValueError(f"world too small: {len(starts)} voxels < M={M}")
    sel = rng.permutation(len(starts))[:M]
    return mean_p[sel].astype(np.float32), mean_n[sel].astype(np.float32)


def make_scan(world: World, N: int, T_gt: np.ndarray, radius: float = 15.0, sigma: float = 0.01, seed: int = 5678):
    """N points sampled within `radius` of the sensor, Gaussian noise along the normal, expressed in the sensor frame."""
    rng = np.random.default_rng(seed)
    c = T_gt[:3, 3]
    got_p, got_n, have = [], [], 0
    while have < N:
        p, n = sample_world(world, max(4 * N, 200000), rng)
        keep = np.linalg.norm(p - c, axis=1) <= radius
        got_p.append(p[keep])
        got_n.append(n[keep])
        have += int(keep.sum())
    p = np.concatenate(got_p)[:N]
    n = np.concatenate(got_n)[:N]
    p = p + n * rng.normal(0, sigma, N)[:, None]
    Tinv = np.linalg.inv(T_gt)
    ps = p @ Tinv[:3, :3].T + Tinv[:3, 3]
    ns = n @ Tinv[:3, :3].T
    return ps.astype(np.float32), ns.astype(np.float32)


def corridor_pose(world: World, k: int, step: float = 0.25, pitch: float = 12.0, x0: float = None) -> np.ndarray:
    """Pose k of a planar trajectory that stays in the aisle between two pillar rows (a ray-cast sensor must not drive
    through a pillar): x advances by `step`, y and yaw weave gently.  `pitch` as given to make_world."""
    L, W, H = world.size
    g = max(int(round(L / pitch)), 1)
    cell = L / g
    yc = -W / 2 + (g // 2 + 1) * cell if g > 1 else 0.25 * W
    x = (-0.4 * L if x0 is None else x0) + step * k
    return make_T(rot_axis_angle([0, 0, 1], 0.25 * math.sin(0.03 * k)), np.array([x, yc + 1.0 * math.sin(0.05 * k), 1.5]))


def loop_pose(world: World, k: int, step: float = 0.25, pitch: float = 12.0, cells=(4, 2), r: float = 2.5) -> np.ndarray:
    """Pose k of a closed planar trajectory: a rectangle of cells[0] x cells[1] grid cells whose sides run along aisle centre
    lines (the pillars sit at cell centres, the aisles at cell boundaries), corners rounded with radius r (inside the clear
    crossing), heading along the direction of travel, arc length k * step from the start (the trajectory simply continues
    into a second lap).  For loop-closure runs: the end of a lap revisits the submap the drive started in."""
    L, W, H = world.size
    g = max(int(round(L / pitch)), 1)
    cell = L / g
    i0, j0 = max((g - cells[0]) // 2, 1), max((g - cells[1]) // 2, 1)
    x0, y0 = -L / 2 + i0 * cell, -W / 2 + j0 * cell
    w, h = cells[0] * cell, cells[1] * cell
    segs = [("line", (x0 + r, y0), 0.0, w - 2 * r), ("arc", (x0 + w - r, y0 + r), -math.pi / 2, None),
            ("line", (x0 + w, y0 + r), math.pi / 2, h - 2 * r), ("arc", (x0 + w - r, y0 + h - r), 0.0, None),
            ("line", (x0 + w - r, y0 + h), math.pi, w - 2 * r), ("arc", (x0 + r, y0 + h - r), math.pi / 2, None),
            ("line", (x0, y0 + h - r), -math.pi / 2, h - 2 * r), ("arc", (x0 + r, y0 + r), math.pi, None)]
    quarter = 0.5 * math.pi * r
    per = 2 * (w - 2 * r) + 2 * (h - 2 * r) + 4 * quarter
    s = (k * step) % per
    for kind, p, a, length in segs:
        ln = length if kind == "line" else quarter
        if s <= ln:
            if kind == "line":
                x, y, yaw = p[0] + s * math.cos(a), p[1] + s * math.sin(a), a
            else:       # counter-clockwise quarter circle about p, starting at angle a
                t = a + s / r
                x, y, yaw = p[0] + r * math.cos(t), p[1] + r * math.sin(t), t + math.pi / 2
            return make_T(rot_axis_angle([0, 0, 1], yaw), np.array([x, y, 1.5]))
        s -= ln
    raise AssertionError("unreachable")


def make_lidar_scan(world: World, T_gt: np.ndarray, beams: int = 64, azimuths: int = 2048, elevation_deg=(-22.5, 22.5),
                    max_range: float = 60.0, sigma: float = 0.01, seed: int = 5678):
    """A spinning-LiDAR sweep ray-cast against the world (64 x 2048 rays, ~130 k returns): one
    ray per (beam, azimuth) cell of a spherical grid, nearest hit among the world's rectangles, Gaussian range noise
    along the ray, returns beyond max_range dropped.  Points and the hit surfaces' normals in the sensor frame."""
    rng = np.random.default_rng(seed)
    el = np.deg2rad(np.linspace(elevation_deg[0], elevation_deg[1], beams))
    az = np.linspace(-math.pi, math.pi, azimuths, endpoint=False)
    ce, se = np.cos(el)[:, None], np.sin(el)[:, None]
    d_s = np.stack([ce * np.cos(az)[None, :], ce * np.sin(az)[None, :], np.broadcast_to(se, (beams, azimuths))], axis=-1).reshape(-1, 3)
    R, o = T_gt[:3, :3], T_gt[:3, 3]
    d = d_s @ R.T
    # only rectangles that can be reached matter
    ext = np.linalg.norm(world.u, axis=1) + np.linalg.norm(world.v, axis=1)
    near = np.nonzero(np.linalg.norm(world.centres - o, axis=1) <= max_range + ext)[0]
    t_best = np.full(d.shape[0], np.inf)
    f_best = np.full(d.shape[0], -1, np.int64)
    for f in near:
        c, u, v, n = world.centres[f], world.u[f], world.v[f], world.normals[f]
        denom = d @ n
        with np.errstate(divide="ignore", invalid="ignore"):
            t = ((c - o) @ n) / denom
        ok = (np.abs(denom) > 1e-12) & (t > 0.05) & (t < t_best)
        if not ok.any():
            continue
        idx = np.nonzero(ok)[0]
        q = o + t[idx, None] * d[idx] - c
        lu, lv = np.linalg.norm(u), np.linalg.norm(v)
        inside = (np.abs(q @ (u / lu)) <= lu) & (np.abs(q @ (v / lv)) <= lv)
        idx = idx[inside]
        t_best[idx] = t[idx]
        f_best[idx] = f
    hit = (f_best >= 0) & (t_best <= max_range)
    r = t_best[hit] + rng.normal(0, sigma, int(hit.sum()))
    ps = d_s[hit] * r[:, None]
    ns = world.normals[f_best[hit]] @ R      # rows: R^T n
    return ps.astype(np.float32), ns.astype(np.float32)


def perturb_pose(T_gt: np.ndarray, trans: float = 0.10, rot_deg: float = 2.0, seed: int = 91011) -> np.ndarray:
    rng = np.random.default_rng(seed)
    d = rng.normal(size=3)
    d /= np.linalg.norm(d)
    ax = rng.normal(size=3)
    dT = make_T(rot_axis_angle(ax, math.radians(rot_deg)), d * trans)
    return T_gt @ dT


@dataclass
class ScanPair:
    map_xyz: np.ndarray
    map_normals: np.ndarray
    scan_xyz: np.ndarray
    scan_normals: np.ndarray
    T_gt: np.ndarray
    T_init: np.ndarray
    voxel: float


def make_scan_pair(N: int, M: int, voxel: float = 0.1, seed: int = 0, radius: float = 15.0, sigma: float = 0.01,
                   trans: float = 0.10, rot_deg: float = 2.0) -> ScanPair:
    """One (scan, map, T_init) triple of SURVEY.md §8(d).  Seeds: 1234+seed (map), 5678+seed (scan), 91011+seed."""
    world = make_world(1.25 * M * voxel * voxel, seed=1234 + seed)
    mp, mn = make_map(world, M, voxel, seed=1234 + seed)
    rng = np.random.default_rng(777 + seed)
    L, W, H = world.size
    lim = max(L / 2 - radius - 1.0, 0.0) if L / 2 > radius + 1 else L / 8
    pos = np.array([rng.uniform(-lim, lim), rng.uniform(-lim, lim), 1.5])
    T_gt = make_T(rot_axis_angle([0, 0, 1], rng.uniform(-math.pi, math.pi)), pos)
    sp, sn = make_scan(world, N, T_gt, radius=min(radius, L / 2), sigma=sigma, seed=5678 + seed)
    T_init = perturb_pose(T_gt, trans, rot_deg, seed=91011 + seed)
    return ScanPair(mp, mn, sp, sn, T_gt, T_init, voxel)
