"""GPU-box helper: the eight per-rank fixtures of `bench.py --gpus 8` (seed = rank) all register and cost the same."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, synthetic as syn
for seed in range(8):
    pair = syn.make_scan_pair(100_000, 2_000_000, 0.1, seed=seed)
    icp = ICP(IcpConfig(use_differential=False, max_iters=50))
    icp.init_reference(pair.map_xyz, pair.map_normals)
    icp.set_reading(pair.scan_xyz, pair.scan_normals)
    T = icp.compute_resident(pair.T_init, with_trace=False)
    dT = np.linalg.inv(pair.T_gt) @ T.astype(np.float64)
    print("seed", seed, "pose err m", float(np.linalg.norm(dT[:3, 3])), "kept", icp.stats.kept_pairs, "gpu_ms", round(icp.stats.gpu_ms, 3))
    icp.close()
