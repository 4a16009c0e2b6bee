"""GPU-box helper: where the first iteration of a call goes (identity T_iter, no incumbents): whole kernel, without the ring
search (dbg 16), and the candidate / row counts of that iteration."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, synthetic as syn
pair = syn.make_scan_pair(100_000, 2_000_000, 0.1, seed=0)
one = ICP(IcpConfig(use_differential=False, max_iters=1, use_graph=False, match_stats=True))
one.init_reference(pair.map_xyz, pair.map_normals)
one.set_reading(pair.scan_xyz, pair.scan_normals)
one.compute_resident(pair.T_init, with_trace=False)
print("first iteration: candidates/query", one.stats.candidates_examined / 1e5, "rows/query", one.stats.cells_probed / 1e5, "matched", one.stats.matched_pairs)
icp = ICP(IcpConfig(use_differential=False, max_iters=20))
icp.init_reference(pair.map_xyz, pair.map_normals)
icp.set_reading(pair.scan_xyz, pair.scan_normals)
icp.compute_resident(pair.T_init)
I = np.eye(4, dtype=np.float32)
# profile_match keeps the incumbents of the previous launch: flag 8 (no outputs) keeps the state "no incumbents" only if it starts so
for flags in (8, 8 | 16, 8 | 2, 8 | 2 | 16):
    fresh = ICP(IcpConfig(use_differential=False, max_iters=1, use_graph=False))
    fresh.init_reference(pair.map_xyz, pair.map_normals)
    fresh.set_reading(pair.scan_xyz, pair.scan_normals)
    fresh.compute_resident(pair.T_init, with_trace=False)   # leaves incumbents behind ...
    import ctypes as C
    from open3d_slam_advanced_rss_2024_public_amd import _lib
    # ... so wipe them: a find_closests on the same cloud resets mq, then set the reading again
    fresh.set_reading(pair.scan_xyz, pair.scan_normals)
    ms = fresh.profile_match(I, 30, flags)
    print(f"flags {flags:2d}: {ms*1e3:7.2f} us  (8 = no outputs so every launch starts without incumbents; 16 = no rings; 2 = no candidates)")
