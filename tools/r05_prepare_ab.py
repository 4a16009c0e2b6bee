"""Device time of prepare_reading (k_read_prep .. k_read_place) on the C2 pair for the library variant in O3S_LIB_VARIANT (timing experiments only: the
variants skip work).  Prints gpu_prepare_us (first kernel of the call -> first matcher launch, the chain's own clock) and the first match / chain."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, synthetic as syn
N = int(os.environ.get("N", "100000")); M = int(os.environ.get("M", "2000000"))
pair = syn.make_scan_pair(N, M, 0.1, seed=0)
y = ICP(IcpConfig())
y.init_reference(pair.map_xyz, pair.map_normals)
y.set_reading(pair.scan_xyz, pair.scan_normals)
prep, ms, gpu = [], [], []
import time
for k in range(43):
    t0 = time.perf_counter(); y.compute_resident(pair.T_init, with_trace=False); dt = time.perf_counter() - t0
    if k >= 3:
        prep.append(y.host_split()[3]); ms.append(1e3 * dt); gpu.append(y.stats.gpu_ms)
print(json.dumps({"variant": os.environ.get("O3S_LIB_VARIANT", "product"), "gpu_prepare_us_median": round(float(np.median(prep)), 2), "ms_per_call_median": round(float(np.median(ms)), 4),
                  "gpu_chain_ms": round(float(np.median(gpu)), 4), "iterations": int(y.stats.iterations)}))
