"""GPU-box helper: time the matcher kernel alone on the converged C2 geometry for a list of debug flags."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, synthetic as syn
N, M = 100_000, 2_000_000
pair = syn.make_scan_pair(N, M, 0.1, seed=0)
cell = float(os.environ.get("CELL", "0"))
icp = ICP(IcpConfig(use_differential=False, max_iters=20, grid_cell=cell))
icp.init_reference(pair.map_xyz, pair.map_normals)
icp.set_reading(pair.scan_xyz, pair.scan_normals)
icp.compute_resident(pair.T_init)
T_conv = icp.stats.trace_T[-1]
flags = [int(x, 0) for x in sys.argv[1:]] or [0]
for f in flags:
    ms = icp.profile_match(T_conv, 100, f)
    print(f"flags={f:#6x}  k_match {ms*1e3:8.2f} us")
ms0 = icp.profile_match(np.eye(4, dtype=np.float32), 20, 0)
print(f"first-iteration pose (identity T_iter): {ms0*1e3:8.2f} us")
