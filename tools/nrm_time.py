"""GPU-box helper: wall time of o3s_scan_preprocess on one ray-cast sweep WITHOUT normals (wide crop, voxel grid, normal estimation,
narrow crop).  KNN (10) and O3S_NRM_RHO (points per occupied cell of the estimator's grid) sweep the estimator.

    KNN=20 python tools/nrm_time.py
"""
import os, sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from open3d_slam_advanced_rss_2024_public_amd import cloud_ops as co, synthetic as syn, ProcessedScan
world = syn.make_world(60000.0, seed=11)
T = syn.corridor_pose(world, 40, 0.25)
sp, sn = syn.make_lidar_scan(world, T, 64, 2048, max_range=60.0, sigma=0.01, seed=340)
sp = sp.astype(np.float64)
ps = ProcessedScan(); ps.set_normal_estimation(float(os.environ.get("RADIUS", "1.0")), int(os.environ.get("KNN", "10")))
wide, narrow = co.croppingVolumeFactory("MaxRadius", 30.0), co.croppingVolumeFactory("MaxRadius", 25.0)
for _ in range(5): ps.preprocess(wide, 0.1, narrow, sp, None)
t0 = time.perf_counter()
for _ in range(20): n = ps.preprocess(wide, 0.1, narrow, sp, None)
print("dbg", os.environ.get("O3S_NRM_DBG", "0"), "knn", os.environ.get("KNN", "10"), "rho", os.environ.get("O3S_NRM_RHO", "-"), "preprocess ms", (time.perf_counter() - t0) / 20 * 1e3, n)
