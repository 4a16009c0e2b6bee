"""-DO3S_TS build: phase stamps (shader cycles, block 0) of k_read_prep on the C2 reading."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, _lib, synthetic as syn
pair = syn.make_scan_pair(100_000, 2_000_000, 0.1, seed=0)
icp = ICP(IcpConfig(use_graph=False))
icp.init_reference(pair.map_xyz, pair.map_normals)
icp.set_reading(pair.scan_xyz, pair.scan_normals)
for _ in range(3):
    icp.compute_resident(pair.T_init)
ts = (C.c_ulonglong * 64)()
assert _lib.lib().o3s_debug_ts(ts) == 0
t = np.array(list(ts), dtype=np.int64)
v = t[56:61]
print("k_read_prep block 0 total", v[-1] - v[0], "cycles:", ", ".join(f"{n}={d}" for n, d in zip(["reset+state+lds-zero", "load+transform+stores", "bin+atomics", "tile flush"], np.diff(v))))
