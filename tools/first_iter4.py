"""GPU-box helper: phases of the far search in a first iteration (timing experiments through the dbg flags of
o3s_icp_profile_match; results of such launches are invalid).  8 = no outputs (every launch starts without incumbents),
16 = no far search at all, 32 = no sweep, 64 = no probe scan, 128 = no probe fetch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, synthetic as syn
cfg = os.environ.get("CFG", "c2")
pair = syn.make_scan_pair(500_000, 20_000_000, 0.02, seed=0) if cfg == "c4" else syn.make_scan_pair(100_000, 2_000_000, 0.1, seed=0)
I = np.eye(4, dtype=np.float32)
fresh = ICP(IcpConfig(use_differential=False, max_iters=1, use_graph=False))
fresh.init_reference(pair.map_xyz, pair.map_normals)
# the pose the first iteration sees = T_refMean^-1 * T_init; profile_match takes T_iter on the PREPARED reading, so prepare with T_init
for flags in (8, 8 | 16, 8 | 32, 8 | 32 | 128, 8 | 32 | 64 | 128, 8 | 64 | 128):
    fresh.set_reading(pair.scan_xyz, pair.scan_normals)
    fresh.compute_resident(pair.T_init, with_trace=False)
    fresh.set_reading(pair.scan_xyz, pair.scan_normals)
    fresh.compute_resident(pair.T_init, with_trace=False)
    # wipe the incumbents the compute left: a dbg-8 launch never writes, but the compute did -> re-prepare through a 1-iteration compute with flag... (the reset happens in prepare)
    ms = fresh.profile_match(I, 20, flags | 0x100)
    print(f"flags {flags:3d}: {ms*1e3:8.2f} us")
