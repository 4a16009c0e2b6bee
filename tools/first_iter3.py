"""GPU-box helper: k_match2 in the FIRST iteration of a call (no incumbents, initial pose 0.1 m / 2 deg off) against a
converged one, HIP events around every launch (o3s_icp_set_profiling).  CFG=c2 (default) | c4 | c5 (0.25 m map, ray cast).

    O3S_FAR=0 python tools/first_iter3.py     # the ring search, for A/B
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, synthetic as syn  # noqa: E402

cfg = os.environ.get("CFG", "c2")
if cfg == "c4":
    pair = syn.make_scan_pair(500_000, 20_000_000, 0.02, seed=0)
elif cfg == "c1":
    pair = syn.make_scan_pair(10_000, 100_000, 0.1, seed=0)
else:
    pair = syn.make_scan_pair(100_000, 2_000_000, 0.1, seed=0)
out = {"cfg": cfg, "far_env": os.environ.get("O3S_FAR", "")}
for iters in (1, 2, 20):
    icp = ICP(IcpConfig(use_differential=False, max_iters=iters, use_graph=False, match_stats=bool(int(os.environ.get("STATS", "0")))))
    icp.init_reference(pair.map_xyz, pair.map_normals)
    icp.set_reading(pair.scan_xyz, pair.scan_normals)
    icp.set_profiling(True)
    ms = []
    for rep in range(5):
        icp.compute_resident(pair.T_init, with_trace=False)
        ms.append(icp.kernel_ms()["match"][0] * 1e3)
    out[f"match_us_avg_over_{iters}_iterations"] = [round(m, 2) for m in ms]
    if iters == 1:
        out["first_iteration_candidates_per_query"] = icp.stats.candidates_examined / len(pair.scan_xyz)
        out["first_iteration_cells_per_query"] = icp.stats.cells_probed / len(pair.scan_xyz)
        out["matched"] = int(icp.stats.matched_pairs)
print(json.dumps(out))
