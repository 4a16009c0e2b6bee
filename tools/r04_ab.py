"""GPU-box helper (round 4 A/B runs; hooks build for the knobs: O3S_LIB_VARIANT=hooks O3S_TAIL=0|1 O3S_FIRST_GROUP=2|4 ...).
Prints one JSON line: the 50-iteration chain (graph replay) per iteration, the icp.yaml chain per registration with the host-side
split (issue / wait / stream queries), the host-buffer call, and k_match2 in the first iteration vs converged (HIP events).
CFG=c2 (default) | c4."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, synthetic as syn  # noqa: E402

cfg = os.environ.get("CFG", "c2")
pair = syn.make_scan_pair(500_000, 20_000_000, 0.02, seed=0) if cfg == "c4" else syn.make_scan_pair(100_000, 2_000_000, 0.1, seed=0)
out = {"cfg": cfg, "env": {k: v for k, v in os.environ.items() if k.startswith("O3S_")}}

icp = ICP(IcpConfig(use_differential=False, max_iters=50))
t0 = time.perf_counter()
icp.init_reference(pair.map_xyz, pair.map_normals)
out["init_reference_ms"] = round(1e3 * (time.perf_counter() - t0), 2)
icp.set_reading(pair.scan_xyz, pair.scan_normals)
for _ in range(4):
    T = icp.compute_resident(pair.T_init, with_trace=False)
reps = 10 if cfg == "c4" else 30
t0 = time.perf_counter()
for _ in range(reps):
    T = icp.compute_resident(pair.T_init, with_trace=False)
dt = time.perf_counter() - t0
out["chain50_ms_per_step"] = round(1e3 * dt / reps, 4)
out["chain50_it_per_s"] = round(50 * reps / dt, 1)
out["chain50_gpu_ms"] = round(icp.stats.gpu_ms, 4)
out["chain50_host_split_us"] = [round(v, 1) for v in icp.host_split()]
dT = np.linalg.inv(pair.T_gt) @ T.astype(np.float64)
out["pose_error_m"] = float(np.linalg.norm(dT[:3, 3]))
# per-kernel (events)
icp.set_profiling(True)
icp.compute_resident(pair.T_init, with_trace=False)
out["kernel_event_us"] = {k: (round(1e3 * v[0], 2), v[1]) for k, v in icp.kernel_ms().items()}
icp.set_profiling(False)
icp.close()

y = ICP(IcpConfig(sort_queries=os.environ.get("SORTQ", "1") == "1"))
y.init_reference(pair.map_xyz, pair.map_normals)
y.set_reading(pair.scan_xyz, pair.scan_normals)
for _ in range(4):
    y.compute_resident(pair.T_init, with_trace=False)
t0 = time.perf_counter()
sp = np.zeros(4)
for _ in range(30):
    y.compute_resident(pair.T_init, with_trace=False)
    sp += np.array(y.host_split())
dt = time.perf_counter() - t0
out["yaml_ms_per_registration"] = round(1e3 * dt / 30, 4)
out["yaml_iterations"] = int(y.stats.iterations)
out["yaml_gpu_chain_ms"] = round(y.stats.gpu_ms, 4)
out["yaml_host_split_us"] = [round(v / 30, 1) for v in sp]
# eager path (no graph)
y2 = ICP(IcpConfig(use_graph=False, sort_queries=os.environ.get("SORTQ", "1") == "1"))
y2.init_reference(pair.map_xyz, pair.map_normals)
y2.set_reading(pair.scan_xyz, pair.scan_normals)
for _ in range(4):
    y2.compute_resident(pair.T_init, with_trace=False)
t0 = time.perf_counter()
sp = np.zeros(4)
for _ in range(30):
    y2.compute_resident(pair.T_init, with_trace=False)
    sp += np.array(y2.host_split())
dt = time.perf_counter() - t0
out["yaml_eager_ms_per_registration"] = round(1e3 * dt / 30, 4)
out["yaml_eager_host_split_us"] = [round(v / 30, 1) for v in sp]
# host buffers handed over every call
t0 = time.perf_counter()
for _ in range(5):
    y.compute(pair.scan_xyz, pair.scan_normals, pair.T_init)
out["yaml_host_buffers_ms_per_call"] = round(1e3 * (time.perf_counter() - t0) / 5, 4)
out["yaml_host_buffers_split_us"] = [round(v, 1) for v in y.host_split()]
y.close()
y2.close()

# first iteration vs converged matcher (events around every launch)
for iters in (1, 20):
    p = ICP(IcpConfig(use_differential=False, max_iters=iters, use_graph=False))
    p.init_reference(pair.map_xyz, pair.map_normals)
    p.set_reading(pair.scan_xyz, pair.scan_normals)
    p.set_profiling(True)
    ms = []
    for _ in range(4):
        p.compute_resident(pair.T_init, with_trace=False)
        ms.append(round(p.kernel_ms()["match"][0] * 1e3, 2))
    out[f"match_us_avg_over_{iters}_iterations"] = ms
    p.close()
print(json.dumps(out))
