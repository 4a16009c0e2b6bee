"""GPU-box helper (hooks build: O3S_LIB_VARIANT=hooks): what the first iteration's k_match2 spends on the row walk and on the
candidates.  profile_match flags: 8 = no outputs (every launch starts without incumbents when combined with 0x100), 16 = no far
search at all (stage 1 only), 32 = far search without candidate loads (bound never tightens: an UPPER bound of the row walk), 64 = the rings' rows are enumerated
and gap-tested but never opened (no window arithmetic, no header loads).
CFG=c2 (default) | c4.  Results of such launches are invalid; the un-flagged numbers come from first_iter3.py."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, synthetic as syn  # noqa: E402

cfg = os.environ.get("CFG", "c2")
pair = syn.make_scan_pair(500_000, 20_000_000, 0.02, seed=0) if cfg == "c4" else syn.make_scan_pair(100_000, 2_000_000, 0.1, seed=0)
I = np.eye(4, dtype=np.float32)
icp = ICP(IcpConfig(use_differential=False, max_iters=1, use_graph=False, match_stats=True))
icp.init_reference(pair.map_xyz, pair.map_normals)
icp.set_reading(pair.scan_xyz, pair.scan_normals)
icp.compute_resident(pair.T_init, with_trace=False)
out = {"cfg": cfg, "first_iteration_candidates_per_query": icp.stats.candidates_examined / len(pair.scan_xyz),
       "first_iteration_ranges_per_query": icp.stats.cells_probed / len(pair.scan_xyz)}
for name, flags in (("full", 8), ("stage1_only", 8 | 16), ("far_rows_only_upper_bound", 8 | 32), ("far_row_enumeration_and_gap_test_only", 8 | 64)):
    icp.set_reading(pair.scan_xyz, pair.scan_normals)
    icp.compute_resident(pair.T_init, with_trace=False)   # prepares the reading under T_init; profile_match then runs with T_iter = I
    out[name + "_us"] = round(icp.profile_match(I, 20, flags | 0x100) * 1e3, 2)
q = (pair.T_init.astype(np.float64) @ np.c_[pair.scan_xyz.astype(np.float64), np.ones(len(pair.scan_xyz))].T).T[:, :3]
idx, d2 = icp.find_closests((q - icp.reference_mean().astype(np.float64)).astype(np.float32))
out["queries_without_a_neighbour_within_max_dist"] = float((idx < 0).mean())
print(json.dumps(out))
