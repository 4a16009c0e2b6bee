import os, sys, time, json
import numpy as np
sys.path.insert(0, os.getcwd())
from open3d_slam_advanced_rss_2024_public_amd import registration as reg, synthetic as syn
world = syn.make_world(20000.0, seed=7)
T = syn.make_T(syn.rot_axis_angle([0, 0, 1], 0.2), np.array([1.0, 2.0, 1.5]))
R, tt = T[:3, :3], T[:3, 3]
tp, tn = syn.make_scan(world, 400000, T, radius=28.0, sigma=0.0, seed=2)
tgt = tp.astype(np.float64) @ R.T + tt; tgt_n = tn.astype(np.float64) @ R.T
src, _ = syn.make_scan(world, 200000, T, radius=25.0, sigma=0.01, seed=3)
src = src.astype(np.float64)
batch = [(src, tgt, tgt_n, syn.perturb_pose(T, 0.15, 2.0, seed=60 + k)) for k in range(16)]
reg.registration_icp_batch(batch[:2], 1.0)
out = {}
for rep in range(2):
    t = time.perf_counter(); reg.registration_icp_batch(batch, 1.0); out[f"batch16_ms_{rep}"] = round(1e3 * (time.perf_counter() - t), 1)
t = time.perf_counter(); [reg.registration_icp(*b[:3], 1.0, b[3]) for b in batch]; out["single16_ms"] = round(1e3 * (time.perf_counter() - t), 1)
from open3d_slam_advanced_rss_2024_public_amd import Submap, cloud_ops as co
big = co.croppingVolumeFactory("MaxRadius", 1.0e6)
A, B = Submap(0.0, big), Submap(0.0, big)
A.setMapPointCloud(src, np.zeros_like(src)); B.setMapPointCloud(tgt, tgt_n)
init = batch[0][3]
reg.registration_icp_submaps(A, B, 1.0, init)
t = time.perf_counter(); rs = [reg.registration_icp_submaps(A, B, 1.0, b[3]) for b in batch]; out["resident16_ms"] = round(1e3 * (time.perf_counter() - t), 1)
out["resident_equals_host"] = bool(np.array_equal(rs[0].transformation, reg.registration_icp(src, tgt, tgt_n, 1.0, init).transformation))
print(os.environ.get("O3S_O3D_LANES"), json.dumps(out))
