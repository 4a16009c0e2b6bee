"""GPU-box helper: duration of the matcher on the FIRST iteration of a call (no incumbents, prior 0.1 m / 2 deg off) and on
a converged one, for the k_match2 variants selected by O3S_GROUP; also the live-use case (icp.yaml chain, new scan)."""
import sys, os, subprocess, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, synthetic as syn
    pair = syn.make_scan_pair(100_000, 2_000_000, 0.1, seed=0)
    icp = ICP(IcpConfig(use_differential=False, max_iters=20))
    icp.init_reference(pair.map_xyz, pair.map_normals)
    icp.set_reading(pair.scan_xyz, pair.scan_normals)
    icp.set_profiling(True)
    icp.compute_resident(pair.T_init)
    icp.set_profiling(False)
    T_conv = icp.stats.trace_T[-1]
    conv = icp.profile_match(T_conv, 100, 0) * 1e3
    # first iteration: one launch at a time from a state without incumbents (a compute() of ONE iteration, events around the kernel)
    one = ICP(IcpConfig(use_differential=False, max_iters=1, use_graph=False))
    one.init_reference(pair.map_xyz, pair.map_normals)
    one.set_reading(pair.scan_xyz, pair.scan_normals)
    one.set_profiling(True)
    firsts = []
    for _ in range(5):
        one.compute_resident(pair.T_init, with_trace=False)
        firsts.append(one.kernel_ms()["match"][0] * 1e3)
    live = ICP(IcpConfig())
    live.init_reference(pair.map_xyz, pair.map_normals)
    t = []
    for _ in range(6):
        t0 = time.perf_counter()
        live.compute(pair.scan_xyz, pair.scan_normals, pair.T_init)
        t.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"env": {k: os.environ[k] for k in ("O3S_GROUP",) if k in os.environ}, "converged_us": round(conv, 2),
                      "first_iteration_us_events": [round(x, 1) for x in firsts], "live_compute_ms": [round(x, 3) for x in t],
                      "live_iterations": live.stats.iterations}))
else:
    for envs in sys.argv[1:]:
        env = dict(os.environ)
        for kv in envs.split():
            k, v = kv.split("=")
            env[k] = v
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-600:], flush=True)
