"""GPU-box helper: the matcher kernel alone on the converged geometry (and on the first-iteration pose) for the
k_match2 variants selected by O3S_GROUP, C2 or C4 (SIZE=c4)."""
import sys, os, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, synthetic as syn
    c4 = os.environ.get("SIZE") == "c4"
    N, M, vox = (500_000, 20_000_000, 0.02) if c4 else (100_000, 2_000_000, 0.1)
    cache = f"/tmp/pair_{N}_{M}.npz"
    if os.path.exists(cache):
        z = np.load(cache); pair = syn.ScanPair(z["a"], z["b"], z["c"], z["d"], z["e"], z["f"], vox)
    else:
        pair = syn.make_scan_pair(N, M, vox, seed=0)
        np.savez(cache, a=pair.map_xyz, b=pair.map_normals, c=pair.scan_xyz, d=pair.scan_normals, e=pair.T_gt, f=pair.T_init)
    icp = ICP(IcpConfig(use_differential=False, max_iters=20))
    icp.init_reference(pair.map_xyz, pair.map_normals)
    icp.set_reading(pair.scan_xyz, pair.scan_normals)
    icp.compute_resident(pair.T_init)
    T_conv = icp.stats.trace_T[-1]
    conv = [icp.profile_match(T_conv, 100, 0) * 1e3 for _ in range(3)]
    nohist = icp.profile_match(T_conv, 100, 1) * 1e3
    import time
    t0 = time.perf_counter()
    for _ in range(10):
        icp.compute_resident(pair.T_init, with_trace=False)
    chain = (time.perf_counter() - t0) / 10 / 20 * 1e6
    print(json.dumps({"env": {k: os.environ[k] for k in ("O3S_GROUP",) if k in os.environ},
                      "converged_us": [round(x, 2) for x in conv], "no_hist_us": round(nohist, 2), "chain_us_per_iter": round(chain, 2)}))
else:
    for envs in sys.argv[1:]:
        env = dict(os.environ)
        for kv in envs.split():
            k, v = kv.split("=")
            env[k] = v
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-400:], flush=True)
