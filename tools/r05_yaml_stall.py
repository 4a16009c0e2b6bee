#!/usr/bin/env python3
"""Per-call record of the icp.yaml chain on the C2 pair (round 5): the driver's line showed a mean of 2.23 ms per
registration where the last call's split added up to 0.30 ms.  Every call's wall time, split, what ended its waits and how the chain
went out — once on a fresh handle, once after the handle churn bench.py's roofline leg does in front of it.

    python tools/r05_yaml_stall.py [--churn] [--calls 40] > gpurun_out/r05_yaml_stall.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig  # noqa: E402
from open3d_slam_advanced_rss_2024_public_amd import synthetic as syn  # noqa: E402


def per_call(icp, T_init, calls):
    rec = []
    for _ in range(calls):
        t0 = time.perf_counter()
        icp.compute_resident(T_init, with_trace=False)
        dt = time.perf_counter() - t0
        sp = icp.host_split_ex()
        sp["ms"] = round(1e3 * dt, 4)
        sp["iterations"] = int(icp.stats.iterations)
        sp["gpu_chain_ms"] = round(icp.stats.gpu_ms, 4)
        rec.append(sp)
    return rec


def summary(rec):
    ms = np.array([r["ms"] for r in rec])
    return {"calls": len(rec), "min": float(ms.min()), "median": float(np.median(ms)), "p99": float(np.percentile(ms, 99)),
            "max": float(ms.max()), "mean": float(ms.mean()),
            "ended_by_post": sum(r["waits_ended_by_post"] for r in rec), "ended_by_event": sum(r["waits_ended_by_event"] for r in rec),
            "ended_by_stream_guard": sum(r["waits_ended_by_stream_guard"] for r in rec),
            "issued": {k: sum(1 for r in rec if r["issued"] == k) for k in ("eager", "captured", "replayed")}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=40)
    ap.add_argument("--churn", action="store_true", help="create / use / destroy the handles bench.py's roofline leg does first")
    a = ap.parse_args()
    pair = syn.make_scan_pair(100_000, 2_000_000, 0.1, seed=0)
    out = {}
    if a.churn:
        icp = ICP(IcpConfig(use_differential=False, max_iters=50), device=0)
        icp.init_reference(pair.map_xyz, pair.map_normals)
        icp.set_reading(pair.scan_xyz, pair.scan_normals)
        for _ in range(25):
            icp.compute_resident(pair.T_init, with_trace=False)
        icp.set_profiling(True)
        for _ in range(3):
            icp.compute_resident(pair.T_init, with_trace=False)
        icp.set_profiling(False)
        for kw in (dict(max_iters=1, use_graph=False), dict(match_stats=True, max_iters=50), dict(match_stats=True, max_iters=1)):
            h = ICP(IcpConfig(use_differential=False, **kw), device=0)
            h.init_reference(pair.map_xyz, pair.map_normals)
            h.set_reading(pair.scan_xyz, pair.scan_normals)
            h.compute_resident(pair.T_init, with_trace=False)
            h.close()
    y = ICP(IcpConfig(), device=0)
    y.init_reference(pair.map_xyz, pair.map_normals)
    y.set_reading(pair.scan_xyz, pair.scan_normals)
    warm = per_call(y, pair.T_init, 3)
    rec = per_call(y, pair.T_init, a.calls)
    out["warmup_calls"] = warm
    out["summary"] = summary(rec)
    out["calls"] = rec
    # back to back, nothing between two calls but a clock read (what round 4's bench loop did)
    ts = [0.0] * 41
    cr = y.compute_resident
    pc = time.perf_counter
    ts[0] = pc()
    for k in range(40):
        cr(pair.T_init, with_trace=False)
        ts[k + 1] = pc()
    out["tight_ms"] = [round(1e3 * (ts[k + 1] - ts[k]), 4) for k in range(40)]
    out["tight_last_split"] = y.host_split_ex()
    # the same with a 50 us pause between calls
    tp = []
    for k in range(40):
        t0 = pc()
        cr(pair.T_init, with_trace=False)
        tp.append(round(1e3 * (pc() - t0), 4))
        t1 = pc()
        while pc() - t1 < 50e-6:
            pass
    out["paused_ms"] = tp
    # the same without graph replay
    y2 = ICP(IcpConfig(use_graph=False), device=0)
    y2.init_reference(pair.map_xyz, pair.map_normals)
    y2.set_reading(pair.scan_xyz, pair.scan_normals)
    per_call(y2, pair.T_init, 3)
    out["eager_summary"] = summary(per_call(y2, pair.T_init, a.calls))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
