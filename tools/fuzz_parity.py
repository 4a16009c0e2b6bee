"""GPU-box helper: randomised differential campaign, HIP path vs CPU oracle, over many small scan/map pairs and chain
configurations.  Every case lands in one class:
  exact   status, iteration count, every per-iteration trim limit and kept count bit-identical, pose within 1e-5
  drift   same status / iteration count / kept counts, limits equal to 1e-5 relative, pose within 1e-5 m / 1e-5 rad
          (the two sides add the same fp64 terms in a different order; when a sum lands within half an fp32 ulp of a
          rounding boundary one pose entry differs by an ulp, and so does every later distance)
  errors_both   both sides stop with the same error status
  diverged      anything else — listed in full, each with `kind`: "far+unbounded" = a third of the scan 50 m away AND maxDist = inf (the
                registration jumps by metres per iteration and amplifies an ulp chaotically), else "other"
exact + drift + errors_both + diverged = cases."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, synthetic as syn
from oracle import oracle as orc

n_cases = int(os.environ.get("CASES", "200"))
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
bad, stats = [], {"cases": 0, "errors_both": 0, "max_dt": 0.0, "max_ang": 0.0, "iters_total": 0}
t0 = time.time()
for case in range(n_cases):
    N = int(rng.integers(200, 6000)); M = int(rng.integers(2000, 60000))
    voxel = float(rng.choice([0.05, 0.1, 0.2]))
    sp = syn.make_scan_pair(N, M, voxel, seed=int(rng.integers(0, 10**6)), trans=float(rng.uniform(0, 0.3)), rot_deg=float(rng.uniform(0, 6)))
    kw = dict(max_dist=float(rng.choice([0.1, 0.3, 0.5, 1.0, np.inf])), trim_ratio=float(rng.choice([-1.0, 0.5, 0.9, 1.0])),
              max_normal_angle=float(rng.choice([-1.0, 0.5, 1.57])), use_differential=bool(rng.integers(0, 2)),
              max_iters=int(rng.integers(1, 25)), smooth_length=int(rng.integers(0, 5)), counter_first=bool(rng.integers(0, 2)))
    gkw = dict(kw)
    for k in ("trim_ratio", "max_normal_angle"):
        if gkw[k] < 0:
            gkw[k] = None
    gkw.update(grid_cell=float(rng.choice([0.0, 0.0, 0.07, 0.31])), sort_queries=bool(rng.integers(0, 2)), use_graph=bool(rng.integers(0, 2)))
    scan = sp.scan_xyz.copy()
    far = rng.random() < 0.1
    if far:
        scan[: N // 3] += 50.0                      # a third of the scan far from the map
    normals = sp.scan_normals if rng.random() < 0.85 else None
    g = ICP(IcpConfig(**gkw)); o = orc.OracleIcp(orc.OracleConfig(**kw), threads=8)
    g.init_reference(sp.map_xyz, sp.map_normals); o.init_reference(sp.map_xyz, sp.map_normals)
    eg = eo = None
    try:
        for _ in range(int(os.environ.get("REPEAT", "1"))):   # REPEAT >= 3: the last call replays a captured graph (use_graph cases)
            Tg = g.compute(scan, normals, sp.T_init)
    except Exception as e:  # noqa: BLE001
        eg = type(e).__name__
    To, code = o.compute(scan, normals, sp.T_init, raise_on_error=False)
    eo = None if code == orc.OK else code
    stats["cases"] += 1
    if stats["cases"] % 25 == 0:   # a silent GPU command is taken to be hung after a few minutes
        print(f"[fuzz] {stats['cases']} / {n_cases} cases, {len(bad)} disagreements, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    rec = dict(case=case, N=N, M=M, far=bool(far), kind=("far+unbounded" if far and not np.isfinite(kw["max_dist"]) else "other"),
               cfg={k: (None if isinstance(v, float) and not np.isfinite(v) else v) for k, v in gkw.items()})
    if (eg is None) != (eo is None):
        bad.append(dict(rec, why="status", gpu=eg, oracle=eo)); continue
    if eg is not None:
        stats["errors_both"] += 1; continue
    stats["iters_total"] += g.stats.iterations
    if g.stats.iterations != o.stats.iterations:
        bad.append(dict(rec, why="iterations", gpu=g.stats.iterations, oracle=o.stats.iterations)); continue
    n = g.stats.iterations
    gl, ol = g.stats.trace_limit[:n], o.trace_limit[:n]
    exact_limits = np.array_equal(gl, ol, equal_nan=True)
    dt, ang = orc.pose_error(To, Tg)
    pose_ok = np.linalg.norm(dt) <= 1e-5 and ang <= 1e-5
    if exact_limits and np.array_equal(g.stats.trace_kept[:n], o.trace_kept[:n]) and pose_ok:
        stats["exact"] = stats.get("exact", 0) + 1
    else:
        fin = np.isfinite(ol)
        close = bool(np.array_equal(np.isfinite(gl), fin) and np.all(np.abs(gl[fin] - ol[fin]) <= 1e-5 * np.abs(ol[fin])))
        first = int(np.argmax(gl != ol)) if not exact_limits else -1
        if close and pose_ok and np.array_equal(g.stats.trace_kept[:n], o.trace_kept[:n]):
            stats["drift"] = stats.get("drift", 0) + 1
        else:
            # where the two trajectories part: the first iteration whose T_iter differs at all, and by how much there.  A first
            # difference of an ulp or two after bit-identical iterations, growing afterwards, is an amplified rounding
            # difference of the fp64 sums' order — not a different correspondence set (tools/fuzz_repro.py shows the matcher side)
            tg, to_ = g.stats.trace_T[:n].astype(np.float64), o.trace_T[:n].astype(np.float64)
            dT = np.abs(tg - to_).reshape(n, -1).max(axis=1) if n else np.zeros(0)
            nz = np.nonzero(dT > 0)[0]
            first_T = int(nz[0]) if nz.size else -1
            kept_eq_before = bool(first_T < 0 or np.array_equal(g.stats.trace_kept[:first_T + 1], o.trace_kept[:first_T + 1]))
            # in ulps of the largest entry of that T_iter (a third of the scan 50 m away makes translations of tens of metres: one
            # ulp there is 1.9e-6)
            ulp = float(np.spacing(np.float32(np.abs(to_[first_T]).max()))) if first_T >= 0 else 1.0
            ulps = float(dT[first_T]) / ulp if first_T >= 0 else 0.0
            amplified = bool(first_T >= 0 and ulps <= 2.0 and kept_eq_before)
            bad.append(dict(rec, why="limits/kept/pose", first_differing_iteration=first, dt=float(np.linalg.norm(dt)), ang=float(ang),
                            first_T_iter_difference_at=first_T, size_of_that_difference=float(dT[first_T]) if first_T >= 0 else 0.0,
                            that_difference_in_ulps_of_the_largest_entry=ulps,
                            kept_counts_equal_up_to_there=kept_eq_before,
                            trajectory_class=("ulp amplified" if amplified else "parts with a visible step")))
    if pose_ok:
        stats["max_dt"] = max(stats["max_dt"], float(np.linalg.norm(dt))); stats["max_ang"] = max(stats["max_ang"], float(ang))
    g.close()
stats["seconds"] = round(time.time() - t0, 1)
stats["diverged"] = len(bad)
stats["diverged_far_unbounded"] = sum(1 for b in bad if b["kind"] == "far+unbounded")
assert stats.get("exact", 0) + stats.get("drift", 0) + stats["errors_both"] + stats["diverged"] == stats["cases"]
print(json.dumps({"stats": stats, "disagreements": bad[:40], "n_disagreements": len(bad)}))
