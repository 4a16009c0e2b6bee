"""GPU-box helper: latency of the Mapper call pattern — icp.yaml chain (<= 15 iterations, Differential checker), host
buffers handed over every call, a different scan size every call (as after voxel down-sampling of live scans)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, synthetic as syn

M = int(os.environ.get("M", 400_000))          # ~ maxNumPoints_ of a submap (Parameters.hpp:106)
pair = syn.make_scan_pair(60_000, M, 0.1, seed=3)
for graph in (True, False):
    icp = ICP(IcpConfig(use_graph=graph, grid_cell=float(os.environ.get('CELL', '0'))))
    t = time.perf_counter(); icp.init_reference(pair.map_xyz, pair.map_normals); t_ref = time.perf_counter() - t
    rng = np.random.default_rng(0)
    lat, its = [], []
    for k in range(40):
        n = int(rng.integers(20_000, 60_000))
        t = time.perf_counter()
        icp.compute(pair.scan_xyz[:n], pair.scan_normals[:n], pair.T_init)
        lat.append(time.perf_counter() - t); its.append(icp.stats.iterations)
    lat = np.array(lat[5:]) * 1e3
    print(f"use_graph={graph}: init_reference({M} pts) {t_ref*1e3:.1f} ms; compute() median {np.median(lat):.2f} ms, p95 {np.percentile(lat,95):.2f} ms, "
          f"iterations median {int(np.median(its))}, gpu chain {icp.stats.gpu_ms:.2f} ms (last)")
