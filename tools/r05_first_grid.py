"""GPU-box helper (round 5): the first-iteration index — a second, coarser grid of the same reference points searched by
iteration 0 of a chain (dense maps only; csrc/o3s_icp.hip init_reference_impl step 4) — against the single grid, on pairs of three
densities.  Hooks build (O3S_FIRST_GRID=0 turns it off, a number sets its cell edge in metres; unset: the library's rule, maxDist / 6); the poses must be bit-equal.
    O3S_LIB_VARIANT=hooks python tools/r05_first_grid.py            # CASES=c4,d03,d05
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, synthetic as syn  # noqa: E402

CASES = {"c4": (500_000, 20_000_000, 0.02), "d03": (300_000, 8_000_000, 0.03), "d04": (250_000, 6_000_000, 0.04), "d05": (200_000, 4_000_000, 0.05), "d07": (150_000, 3_000_000, 0.07),
         "c2": (100_000, 2_000_000, 0.1)}
for name in os.environ.get("CASES", "c4,d03,d05").split(","):
    n, m, voxel = CASES[name]
    pair = syn.make_scan_pair(n, m, voxel, seed=0)
    ref = {}
    for edge in os.environ.get("EDGES", "0,default,0.07,0.0833,0.095").split(","):
        if edge == "default":
            os.environ.pop("O3S_FIRST_GRID", None)
        else:
            os.environ["O3S_FIRST_GRID"] = edge
        row = {"case": name, "scan": n, "map": m, "voxel": voxel, "first_grid_edge_m": edge}
        for label, kw in (("fixed50", dict(use_differential=False, max_iters=50)), ("yaml", dict(use_differential=True, max_iters=15))):
            icp = ICP(IcpConfig(**kw))
            if label == "fixed50":
                os.environ["O3S_PRINT_GRID"] = "1"
            else:
                os.environ.pop("O3S_PRINT_GRID", None)
            t0 = time.perf_counter()
            icp.init_reference(pair.map_xyz, pair.map_normals)
            row["init_reference_s"] = round(time.perf_counter() - t0, 3)
            icp.set_reading(pair.scan_xyz, pair.scan_normals)
            for _ in range(3):
                T = icp.compute_resident(pair.T_init, with_trace=False)
            ts = []
            for _ in range(6):
                t0 = time.perf_counter()
                T = icp.compute_resident(pair.T_init, with_trace=False)
                ts.append(time.perf_counter() - t0)
            it = int(icp.stats.iterations)
            row[label] = {"ms_per_call_median": round(1e3 * float(np.median(ts)), 4), "iterations": it, "gpu_chain_ms": round(float(icp.stats.gpu_ms), 4)}
            if label == "fixed50":
                row[label]["it_per_s"] = round(it / float(np.median(ts)), 1)
            key = (label,)
            if key not in ref:
                ref[key] = np.asarray(T).copy()
            row[label]["pose_bit_equal_to_single_grid"] = bool(np.array_equal(np.asarray(T), ref[key]))
            icp.close()
        # the first iteration's matcher launch by itself (events around every launch of an eagerly issued one-iteration chain)
        os.environ.pop("O3S_PRINT_GRID", None)
        icp = ICP(IcpConfig(use_differential=False, max_iters=1, use_graph=False))
        icp.init_reference(pair.map_xyz, pair.map_normals)
        icp.set_reading(pair.scan_xyz, pair.scan_normals)
        icp.set_profiling(True)
        ms = []
        for _ in range(4):
            icp.compute_resident(pair.T_init, with_trace=False)
            ms.append(icp.kernel_ms()["match"][0] * 1e3)
        row["first_iteration_match_us"] = round(min(ms), 1)
        icp.close()
        print(json.dumps(row), flush=True)
