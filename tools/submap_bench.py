"""GPU-box helper: per-scan cost of the device-resident submap (insert + re-voxelise, crop + reference hand-over) next
to the same steps through the CPU oracle's host loops (single thread, like the reference)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, Submap, cloud_ops as co, synthetic as syn
from oracle import oracle as orc

n_scans = int(os.environ.get("SCANS", "12"))
n_pts = int(os.environ.get("PTS", "100000"))
voxel = float(os.environ.get("VOXEL", "0.1"))
world = syn.make_world(40000.0, seed=7)
traj = []
for k in range(n_scans):
    T = syn.make_T(syn.rot_axis_angle([0, 0, 1], 0.05 * k), np.array([-12.0 + 2.0 * k, 0.5 * k, 1.5]))
    sp, sn = syn.make_scan(world, n_pts, T, radius=20.0, sigma=0.01, seed=200 + k)
    traj.append((sp.astype(np.float64), sn.astype(np.float64), T))

sm = Submap(voxel, co.croppingVolumeFactory("MaxRadius", 25.0))
icp = ICP(IcpConfig())
t_ins, t_ref, sizes, patches = [], [], [], []
for sp, sn, T in traj:
    t0 = time.perf_counter(); sm.insertScan(sp, sn, T); t_ins.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); k = sm.set_reference(co.croppingVolumeFactory("MaxRadius", 20.0), T, icp); t_ref.append(time.perf_counter() - t0)
    sizes.append(len(sm)); patches.append(k)

# CPU: the oracle's restatement of the same host loops on the same inputs (last 3 scans only: it is slow)
mp, mn = None, None
c_ins, c_ref = [], []
for i, (sp, sn, T) in enumerate(traj):
    t0 = time.perf_counter()
    tp, tn = orc.transform_cloud(T, sp, sn)
    p = tp if mp is None else np.concatenate([mp, tp]); n = tn if mn is None else np.concatenate([mn, tn])
    mp, mn, _ = orc.voxelize_within_crop(orc.make_cropper("MaxRadius", 25.0, centre=T[:3, 3]), voxel, p, n)
    c_ins.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    mask = orc.crop_mask(orc.make_cropper("MaxRadius", 20.0, centre=T[:3, 3]), mp)
    xyzw, n32 = orc.o3d_to_pm(mp[mask], mn[mask])
    o = orc.OracleIcp(orc.OracleConfig(), threads=1); o.init_reference(xyzw[:, :3], n32)
    c_ref.append(time.perf_counter() - t0)
print(json.dumps({"scans": n_scans, "scan_points": n_pts, "voxel": voxel, "map_points_final": sizes[-1], "patch_points_final": patches[-1],
                  "gpu_insert_ms_median": round(1e3 * float(np.median(t_ins[2:])), 3), "gpu_set_reference_ms_median": round(1e3 * float(np.median(t_ref[2:])), 3),
                  "cpu_insert_ms_median": round(1e3 * float(np.median(c_ins[2:])), 3), "cpu_crop_convert_initref_ms_median": round(1e3 * float(np.median(c_ref[2:])), 3),
                  "map_size_equal_cpu": bool(sizes[-1] == mp.shape[0])}))
