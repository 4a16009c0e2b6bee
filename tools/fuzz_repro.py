"""Replays tools/fuzz_parity.py up to one case and compares the matcher outputs of GPU and oracle per iteration."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, synthetic as syn
from oracle import oracle as orc
target = int(sys.argv[1]); rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
for case in range(target + 1):
    N = int(rng.integers(200, 6000)); M = int(rng.integers(2000, 60000))
    voxel = float(rng.choice([0.05, 0.1, 0.2]))
    seed = int(rng.integers(0, 10**6)); trans = float(rng.uniform(0, 0.3)); rot = float(rng.uniform(0, 6))
    kw = dict(max_dist=float(rng.choice([0.1, 0.3, 0.5, 1.0, np.inf])), trim_ratio=float(rng.choice([-1.0, 0.5, 0.9, 1.0])),
              max_normal_angle=float(rng.choice([-1.0, 0.5, 1.57])), use_differential=bool(rng.integers(0, 2)),
              max_iters=int(rng.integers(1, 25)), smooth_length=int(rng.integers(0, 5)), counter_first=bool(rng.integers(0, 2)))
    gkw = dict(kw)
    for k in ("trim_ratio", "max_normal_angle"):
        if gkw[k] < 0:
            gkw[k] = None
    gkw.update(grid_cell=float(rng.choice([0.0, 0.0, 0.07, 0.31])), sort_queries=bool(rng.integers(0, 2)), use_graph=bool(rng.integers(0, 2)))
    far = rng.random() < 0.1
    has_n = rng.random() < 0.85
    if case < target:
        continue
    sp = syn.make_scan_pair(N, M, voxel, seed=seed, trans=trans, rot_deg=rot)
    scan = sp.scan_xyz.copy()
    if far:
        scan[: N // 3] += 50.0
    normals = sp.scan_normals if has_n else None
    print("case", case, "N", N, "M", M, "voxel", voxel, "far", far, "normals", has_n, gkw)
    g = ICP(IcpConfig(**gkw)); o = orc.OracleIcp(orc.OracleConfig(**kw), threads=8)
    g.init_reference(sp.map_xyz, sp.map_normals); o.init_reference(sp.map_xyz, sp.map_normals)
    Tg = g.compute(scan, normals, sp.T_init); To, code = o.compute(scan, normals, sp.T_init, raise_on_error=False)
    n = min(g.stats.iterations, o.stats.iterations)
    print("iters", g.stats.iterations, o.stats.iterations)
    print("limits gpu", g.stats.trace_limit[:n]); print("limits orc", o.trace_limit[:n])
    print("kept gpu", g.stats.trace_kept[:n]); print("kept orc", o.trace_kept[:n])
    print("max |T_iter(gpu) - T_iter(oracle)| per iteration:", [float(np.abs(g.stats.trace_T[k].astype(np.float64) - o.trace_T[k].astype(np.float64)).max()) for k in range(n)])
    Tc = np.eye(4); Tc[:3, 3] = -g.reference_mean().astype(np.float64)
    for it in range(n):
        Ti = o.trace_T[it - 1].astype(np.float64) if it > 0 else np.eye(4)
        q = orc.rigid_transform((Ti.astype(np.float32) @ (Tc @ sp.T_init).astype(np.float32)), scan)[0]
        gi, gd = g.find_closests(q); oi, od = o.find_closests(q)
        bad = np.nonzero((gi != oi) | (gd != od))[0]
        print("iter", it, "matcher mismatches", bad.size, [(int(b), int(gi[b]), int(oi[b]), float(gd[b]), float(od[b])) for b in bad[:5]])
        if bad.size:
            b = bad[0]
            print("   query", q[b], "gpu ref", (sp.map_xyz[gi[b]] - g.reference_mean()) if gi[b] >= 0 else None, "orc ref", (sp.map_xyz[oi[b]] - g.reference_mean()) if oi[b] >= 0 else None)
            break
