"""GPU-box helper (round 5): what would a SECOND, coarser grid consulted only by the first iteration of a call buy?
An upper bound without building it: the matcher's cell edge is a parameter (IcpConfig.grid_cell; results do not depend on it), so the
first iteration (no incumbents: the far search does the work) and a converged one are timed over a sweep of cell edges on the same
pair.  A two-grid layout can at best run the first iteration at the best edge of column one and the rest at the best edge of column two.
CFG=c2 (default) | c4.  HIP events around every launch (o3s_icp_set_profiling), match_stats for the counts.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402,F401

from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, synthetic as syn  # noqa: E402

cfg = os.environ.get("CFG", "c2")
if cfg == "c4":
    pair = syn.make_scan_pair(500_000, 20_000_000, 0.02, seed=0)
    cells = [0.0, 0.045, 0.07, 0.085, 0.114, 0.17, 0.25]
else:
    pair = syn.make_scan_pair(100_000, 2_000_000, 0.1, seed=0)
    cells = [0.0, 0.17, 0.25, 0.4, 0.5, 0.7, 1.0]
if os.environ.get("CELLS"):
    cells = [float(v) for v in os.environ["CELLS"].split(",")]
rows = []
T_ref = None
for cell in cells:
    row = {"grid_cell": cell}
    for iters in (1, 20):
        icp = ICP(IcpConfig(use_differential=False, max_iters=iters, use_graph=False, grid_cell=cell))
        icp.init_reference(pair.map_xyz, pair.map_normals)
        icp.set_reading(pair.scan_xyz, pair.scan_normals)
        icp.set_profiling(True)
        ms = []
        for rep in range(4):
            T = icp.compute_resident(pair.T_init, with_trace=False)
            ms.append(icp.kernel_ms()["match"][0] * 1e3)
        if iters == 1:
            row["first_iteration_match_us"] = round(min(ms), 2)
        else:
            first = row["first_iteration_match_us"]
            row["converged_match_us"] = round((min(ms) * iters - first) / (iters - 1), 2)
            if T_ref is None:
                T_ref = np.asarray(T).copy()
            row["pose_equals_the_default_cell_s"] = bool(np.array_equal(np.asarray(T), T_ref))
        icp.close() if hasattr(icp, "close") else None
    icp = ICP(IcpConfig(use_differential=False, max_iters=1, use_graph=False, match_stats=True, grid_cell=cell))   # the counted run (slower kernel: apart)
    icp.init_reference(pair.map_xyz, pair.map_normals)
    icp.set_reading(pair.scan_xyz, pair.scan_normals)
    icp.compute_resident(pair.T_init, with_trace=False)
    row["first_iteration_candidates_per_query"] = round(icp.stats.candidates_examined / len(pair.scan_xyz), 1)
    row["first_iteration_cells_per_query"] = round(icp.stats.cells_probed / len(pair.scan_xyz), 1)
    icp.close() if hasattr(icp, "close") else None
    rows.append(row)
    print(json.dumps(row), flush=True)
print(json.dumps({"cfg": cfg, "rows": rows}))
