#!/usr/bin/env python3
"""bench.py — ICP iterations/s of the MI355X scan-to-map ICP path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one compute() of the hot path over one (scan, map) pair already resident in HBM: transform + spatial sort
of the 100k-pt scan, then 50 fixed ICP iterations (match, trim select, centroid, normal equations, solve) against the
2M-pt map index, then the 4x4 pose back to the host.  Workload = BASELINE.json configs[1] (C2).  With N > 1 every rank
owns one GPU and an independent (scan, map) pair (different seed): no data-path collective, weak scaling.

Rank 0 prints ONE JSON line.  `roofline` prices the matcher kernel (k_match2) with the algorithmic-byte model of
SURVEY.md §8(d) / DESIGN.md and its average launch duration MEASURED IN THIS RUN: HIP events on the library's stream around
every launch of a timed chain, minus what an event pair measures around an empty kernel (o3s_icp_event_gap_ms); the
rocprofv3 --kernel-trace average committed under profiles/ rides along as `profiled_avg_ms`, and a disagreement beyond 15 %
is reported on stderr and in the record (`--strict`: exit code 3).  `roofline_kernels` does the same for every kernel of
the chain.  `cpu_baseline` times the CPU oracle (oracle/, the checker — never the thing shipped) on the same inputs on this
box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


class timed_region:
    """Holds the interpreter's garbage collector off while a loop is being timed and records every collection that still runs.
    With torch imported a generation-2 collection takes ~40 ms; round 4's driver line had one land in call 19 of the twenty
    0.27 ms icp.yaml registrations (mean 2.2 ms; profiles/r05/a_yaml_stall_probe.txt).  It is the harness's, not the path's — a C++
    host has none — so the loops are timed with the collector emptied first and switched off.  O3S_BENCH_KEEP_GC=1 leaves it on."""

    def __init__(self):
        self.collections = []   # (generation, ms)
        self._t = 0.0

    def _cb(self, phase, info):
        if phase == "start":
            self._t = time.perf_counter()
        else:
            self.collections.append((int(info.get("generation", -1)), round(1e3 * (time.perf_counter() - self._t), 3)))

    def __enter__(self):
        import gc

        self._gc = gc
        self._was = gc.isenabled()
        self.keep = os.environ.get("O3S_BENCH_KEEP_GC") == "1"
        gc.collect()
        gc.callbacks.append(self._cb)
        if not self.keep:
            gc.disable()
        return self

    def __exit__(self, *exc):
        self._gc.callbacks.remove(self._cb)
        if self._was:
            self._gc.enable()
        return False

    def record(self):
        return {"collector_on_while_timed": self.keep, "collections_while_timed": self.collections}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scan", type=int, default=100_000, help="reading points N (C2: 100k)")
    ap.add_argument("--map", type=int, default=2_000_000, help="reference points M (C2: 2M)")
    ap.add_argument("--voxel", type=float, default=0.1)
    ap.add_argument("--iters", type=int, default=50, help="fixed ICP iterations per compute (C2: 50)")
    ap.add_argument("--grid-cell", type=float, default=0.0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-iters", type=int, default=100, help="iterations of the CPU sample (100: about 15 core-seconds on the box's 16-core share)")
    ap.add_argument("--no-sort", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--mode", choices=["pairs", "sharded"], default="pairs",
                    help="pairs (default): independent pairs, --pairs-per-gpu of them per GPU, weak scaling, no data-path collective; "
                         "sharded: ONE pair, the scan split over the ranks with three small all-reduces per iteration (strong scaling, "
                         "SURVEY.md 8(e) mode 2)")
    ap.add_argument("--pairs-per-gpu", type=int, default=1,
                    help="pairs mode: independent (scan, map) pairs every rank keeps in flight through o3s_icp_compute_batch "
                         "(8 with --gpus 8 = BASELINE config 3's 64 pairs); value = iterations/s summed over all pairs")
    ap.add_argument("--strict", action="store_true", help="exit with code 3 when the live kernel time and the committed rocprofv3 average differ by more than 15 %")
    ap.add_argument("--exchange", choices=["rccl", "torch"], default="rccl",
                    help="sharded mode: ncclAllReduce issued from C (libo3dslam_icp_rccl.so) or dist.all_reduce from Python")
    ap.add_argument("--batch-pairs", type=int, default=8, help="pairs kept in flight on one GPU for extra.batched_on_one_gpu (0/1: skip)")
    ap.add_argument("--no-c4", action="store_true", help="skip extra.c4 (BASELINE config 4: 500k-pt scan vs 20M-pt map at 0.02 m; ~40 s of fixture generation)")
    ap.add_argument("--no-c3", action="store_true", help="skip extra.c3 (BASELINE config 3: 64 pairs of 100k vs 400k on this GPU; ~20 s of fixture generation)")
    ap.add_argument("--no-c5", action="store_true", help="skip extra.c5 (BASELINE config 5: the per-scan loop through the compiled driver; ~70 s, most of it ray casting)")
    ap.add_argument("--no-sharded-extra", action="store_true", help="N > 1: skip extra.sharded_one_pair (the one-pair-sharded mode on the same ranks)")
    ap.add_argument("--timing-only", action="store_true", help="only the timed region (for rocprofv3 runs): no roofline / PCIe / CPU legs")
    return ap.parse_args()


def sharded_roofline(icp, make_sharded, pair, n_rank, iters):
    """The roofline object of a one-pair-sharded line, per RANK: the matcher over the rank's slice of the reading (n_rank points)
    against the replicated index.  Duration: the matcher alone on the resident slice at the chain's converged pose, HIP events on
    the library's stream around 50 launches (o3s_icp_profile_match — a sharded chain cannot be bracketed launch by launch, its
    collectives have to stay matched across ranks); bytes: the converged launches' candidates per query, counted by a sharded
    chain with match_stats (all - first) / (iterations - 1).  Every rank calls this (the counted chains carry collectives)."""
    icp.compute_resident(pair.T_init, with_trace=True)
    T_last = icp.stats.trace_T[-1]
    conv_ms = icp.profile_match(T_last, reps=50)
    st_all = make_sharded(dict(match_stats=True))
    st_all.compute_resident(pair.T_init, with_trace=False)
    c_all = st_all.stats.candidates_examined / (n_rank * iters)
    st_all.close()
    st_1 = make_sharded(dict(match_stats=True, max_iters=1))
    st_1.compute_resident(pair.T_init, with_trace=False)
    c_first = st_1.stats.candidates_examined / n_rank
    st_1.close()
    c_conv = (c_all * iters - c_first) / max(iters - 1, 1)
    nbytes = n_rank * (236.0 + 12.0 * c_conv)
    ach = nbytes / (conv_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "k_match2", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
            "traffic": None, "per": "rank", "points_per_rank": int(n_rank), "alg_bytes_per_launch": int(nbytes), "avg_launch_ms": round(conv_ms, 5),
            "avg_launch_ms_source": "measured in this run on this rank: HIP events on the library's stream around 50 launches of the matcher alone over the "
                                    "rank's resident slice at the chain's converged pose (o3s_icp_profile_match)",
            "cbar_candidates_per_query": {"all": round(c_all, 2), "first": round(c_first, 2), "converged": round(c_conv, 2)},
            "binding_limit": "instruction issue + dependent cache round trips (a slice of n / world points is less than one generation of waves)"}


def run_sharded(args, rank, world, device, dist, torch):
    """ONE (scan, map) pair over all ranks: value = iterations/s of that single registration (strong scaling)."""
    import ctypes as C

    from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, _lib
    from open3d_slam_advanced_rss_2024_public_amd import synthetic as syn
    from open3d_slam_advanced_rss_2024_public_amd.parallel import PairSharded, shard_slice

    N, M, iters = args.scan, args.map, args.iters
    pair = syn.make_scan_pair(N, M, args.voxel, seed=0)  # the same pair on every rank
    cfg = IcpConfig(use_differential=False, max_iters=iters, grid_cell=args.grid_cell, sort_queries=not args.no_sort)
    own_group = False
    if dist is None:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", device))
        own_group = True
    comm = None
    if args.exchange == "rccl":
        R = _lib.rccl_lib()
        uid = C.create_string_buffer(128)
        if rank == 0:
            assert R.o3s_rccl_unique_id(uid) == 0, R.o3s_rccl_last_error()
        t = torch.frombuffer(bytearray(uid.raw), dtype=torch.uint8).to(f"cuda:{device}")
        dist.broadcast(t, src=0)
        uid = C.create_string_buffer(bytes(t.cpu().numpy().tobytes()), 128)
        comm = C.c_void_p()
        assert R.o3s_rccl_create(uid, rank, world, device, C.byref(comm)) == 0, R.o3s_rccl_last_error()
        icp = ICP(cfg, device=device)
        assert icp.init_reference(pair.map_xyz, pair.map_normals)
        sl = shard_slice(N, world, rank)
        icp.set_reading(pair.scan_xyz[sl], pair.scan_normals[sl])
        icp.shard_configure_rccl(N, rank, world, comm.value)
        run = lambda: icp.compute_resident(pair.T_init, with_trace=False)  # noqa: E731

        def make_sharded(kw):
            c = IcpConfig(use_differential=False, max_iters=iters, grid_cell=args.grid_cell, sort_queries=not args.no_sort)
            for k_, v_ in kw.items():
                setattr(c, k_, v_)
            h_ = ICP(c, device=device)
            assert h_.init_reference(pair.map_xyz, pair.map_normals)
            h_.set_reading(pair.scan_xyz[sl], pair.scan_normals[sl])
            h_.shard_configure_rccl(N, rank, world, comm.value)
            return h_
    else:
        ps = PairSharded(cfg, device=device)
        assert ps.init_reference(pair.map_xyz, pair.map_normals)
        ps.set_reading(pair.scan_xyz, pair.scan_normals)
        run = lambda: ps.compute(pair.T_init, with_trace=False)  # noqa: E731
    for _ in range(args.warmup):
        run()
    dist.barrier()
    torch.cuda.synchronize()
    with timed_region():
        t0 = time.perf_counter()
        for _ in range(args.steps):
            T = run()
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
    t = torch.tensor([t1 - t0], dtype=torch.float64, device=f"cuda:{device}")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    roofline = None
    if args.exchange == "rccl":   # every rank: the counted chains carry collectives
        roofline = sharded_roofline(icp, make_sharded, pair, sl.stop - sl.start, iters)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline_record(args, pair, lambda **kw: IcpConfig(**dict(dict(use_differential=False, max_iters=iters), **kw)), device)
    line = None
    if rank == 0:
        dT = np.linalg.inv(pair.T_gt) @ T.astype(np.float64)
        line = json.dumps({
            "metric": _metric_name(args.scan, args.map), "value": round(iters * args.steps / elapsed, 2),
            "unit": "ICP iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"ONE pair sharded: {N}-pt scan split over {world} rank(s) vs replicated {M}-pt voxel map, "
                                   f"{args.voxel} m voxels, {iters} iters, icp.yaml chain",
                       "scan_points": N, "map_points": M, "iterations_per_step": iters,
                       "parallelism": f"reading split {world}-way, 3 in-place sum all-reduces/iteration ({args.exchange}): int32 x (16 >> floor(log2(world))) x 2048 "
                                      "(level-1 histogram replicas), int32 x 8192 (level 2, 13 bits), f64 x (128 + 34 x 128 + 34 x blocks) (level-3 counts, "
                                      "per-bin and per-block raw moments of the kept pairs: centred algebraically after the exchange)"},
            "roofline": roofline, "cpu_baseline": cpu,
            "extra": {"pose_error_vs_ground_truth_m": float(np.linalg.norm(dT[:3, 3])), "rccl_ranks": world if args.exchange == "rccl" else 0,
                      "rccl_collectives_total": int(R.o3s_rccl_collectives(comm)) if args.exchange == "rccl" else None}})
    if own_group or world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return line


def roofline_of(icp, cfg, pair, N, M, voxel, iters, it_per_s_one_pair, gpu_ms_chain, device):
    """The roofline objects of one workload (C2 for the headline, C4 for extra.c4): every kernel of the chain priced with a
    duration measured in THIS run.
    (i)   in-chain durations: the same chain issued eagerly with a HIP event recorded on the library's stream between every two
          launches (o3s_icp_set_profiling), averaged over ALL iterations of three steps.  An event pair brackets the kernel plus
          a dispatch gap and the event itself; that per-launch overhead g is measured in the run too: the timed steps replay the
          SAME chain as a graph and the chain stamps its own clock (stats.gpu_ms), so
          sum_k event_k = chain_ms / iterations + n_kernels * g gives g, and kernel_k = event_k - g.
    (ii)  the matcher's FIRST iteration (no incumbents: the far search does the work — and it is most of what the 3-5-iteration
          chain of icp.yaml runs) apart from the converged ones: duration from a one-iteration chain, candidates per query from
          a one-iteration counted run; converged = (all - first) / (iterations - 1) on BOTH sides, so each fraction divides the
          bytes of the launches by the time of the same launches."""
    from open3d_slam_advanced_rss_2024_public_amd import ICP

    icp.set_profiling(True)
    acc = {}
    for _ in range(3):
        icp.compute_resident(pair.T_init, with_trace=False)
        for k_, (ms_, n_) in icp.kernel_ms().items():
            a_ = acc.setdefault(k_, [0.0, 0])
            a_[0] += ms_ * n_
            a_[1] += n_
    icp.set_profiling(False)
    kms = {k_: ((v_[0] / v_[1]) if v_[1] else 0.0, v_[1]) for k_, v_ in acc.items()}
    fused = kms["normal_eq"][1] == 0   # up to 131 k points selection + normal equations are one launch (k_sel_ne), timed under "sel_finish"
    has_solve = kms["solve"][1] > 0    # a k_solve launch closes the iteration (always, unless k_sel_ne carries the closing tail)
    icp.compute_resident(pair.T_init, with_trace=False)  # restore the resident state (graph replay) after the eager profile
    # first iteration alone: a one-iteration chain, events around its launches (three calls, the last two averaged)
    icp1 = ICP(cfg(max_iters=1, use_graph=False), device=device)
    icp1.init_reference(pair.map_xyz, pair.map_normals)
    icp1.set_reading(pair.scan_xyz, pair.scan_normals)
    icp1.set_profiling(True)
    first_ev = []
    for _ in range(3):
        icp1.compute_resident(pair.T_init, with_trace=False)
        first_ev.append(icp1.kernel_ms()["match"][0])
    icp1.close()
    first_ev_ms = float(np.mean(first_ev[1:]))
    # candidates per query: all iterations, and the first one alone (counted runs)
    st_all = ICP(cfg(match_stats=True), device=device)
    st_all.init_reference(pair.map_xyz, pair.map_normals)
    st_all.set_reading(pair.scan_xyz, pair.scan_normals)
    st_all.compute_resident(pair.T_init, with_trace=False)
    cbar = st_all.stats.candidates_examined / (N * iters)
    rows = st_all.stats.cells_probed / (N * iters)
    matched = int(st_all.stats.matched_pairs)
    st_all.close()
    st_1 = ICP(cfg(match_stats=True, max_iters=1), device=device)
    st_1.init_reference(pair.map_xyz, pair.map_normals)
    st_1.set_reading(pair.scan_xyz, pair.scan_normals)
    st_1.compute_resident(pair.T_init, with_trace=False)
    cbar_first = st_1.stats.candidates_examined / N
    st_1.close()
    cbar_conv = (cbar * iters - cbar_first) / max(iters - 1, 1)
    # committed rocprofv3 evidence for this workload (kernel-trace averages, PMC traffic): newest round first
    workload_tag = "c2" if (N, M) == (100_000, 2_000_000) and voxel == 0.1 else \
                   "c4" if (N, M) == (500_000, 20_000_000) and voxel == 0.02 else None
    prof = None
    if workload_tag:
        for rnd in ("r05", "r04", "r03", "r02"):
            pf = os.path.join(ROOT, "profiles", rnd, f"roofline_inputs_{workload_tag}.json")
            if os.path.exists(pf):
                with open(pf) as f:
                    prof = json.load(f)
                prof["_file"] = os.path.relpath(pf, ROOT)
                break
    prof_kernels = (prof or {}).get("kernels_avg_us", {})
    if prof and "k_match2" not in prof_kernels and prof.get("k_match_avg_us"):
        prof_kernels = dict(prof_kernels, k_match2=prof["k_match_avg_us"])
    traffic = int(prof["hbm_bytes_per_launch"]) if prof and prof.get("hbm_bytes_per_launch") else None
    traffic_src = prof.get("source") if prof else None
    # algorithmic bytes per launch (per reading point unless stated):
    #   k_match2   12 xyz stream + 216 = 27 cell headers x 8 + 12 per candidate examined + 8 (d2, slot) written   
    #   k_classify 12 + 12 (xyz, normal) + 8 (d2, slot) + 16 (matched point) + 16 (normal gather) + 16 (normal out), + 32 per
    #              undecided pair (the trim bin: ~2 % of the matched pairs)
    #   k_sel_ne   20 (xyz, d2, slot) + 32 (matched point, normal) per point, + 32 per undecided pair for the selection sweep
    #              (two kernels beyond 131 k points: k_sel_finish 32 per undecided pair, k_normal_eq 52 per point)
    #   k_solve    27 x 8 per block partial + the state in and out: a single-block dependency chain, not a stream
    undecided = 0.02 * matched
    nb_part = min(512, -(-N // 512))
    alg = {"k_match2": N * (236.0 + 12.0 * cbar), "k_classify": N * 80.0 + 32.0 * undecided}
    if has_solve:
        alg["k_solve"] = 27.0 * 8.0 * nb_part + 2 * 1008.0
    if fused:
        alg["k_sel_ne"] = N * 52.0 + 32.0 * undecided
    else:
        # beyond 262 k points (more classify blocks than the finishing block has threads) the candidate sweep runs on many blocks
        # first: k_sel_partial + k_sel_finish share one event bracket
        sel_name = "k_sel_partial+k_sel_finish" if -(-N // 512) > 512 else "k_sel_finish"
        alg[sel_name] = 32.0 * undecided + 7 * 8.0 * (-(-N // 512))
        alg["k_normal_eq"] = N * 52.0
    chain_iter_ms = gpu_ms_chain / iters  # one iteration of the graph-replayed chain of the timed steps (the chain's own clock)
    event_of = {"k_match2": "match", "k_classify": "classify", "k_sel_ne": "sel_finish", "k_sel_finish": "sel_finish", "k_sel_partial+k_sel_finish": "sel_finish",
                "k_normal_eq": "normal_eq", "k_solve": "solve"}
    gap_ms = max((sum(kms[event_of[n_]][0] for n_ in alg) - chain_iter_ms) / len(alg), 0.0)
    kernels = []
    disagree = []
    for name, nbytes in alg.items():
        ev_ms = kms[event_of[name]][0]
        live_ms = max(ev_ms - gap_ms, 1e-6)
        p_us = sum(prof_kernels.get(n_, 0.0) for n_ in name.split("+")) if all(n_ in prof_kernels for n_ in name.split("+")) else None
        ent = {"kernel": name, "alg_bytes_per_launch": int(nbytes), "avg_launch_ms": round(live_ms, 5), "event_pair_ms": round(ev_ms, 5),
               "achieved": round(nbytes / (live_ms * 1e-3) / 1e9, 2), "unit": "GB/s",
               "frac": round(nbytes / (live_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
               "profiled_avg_ms": round(p_us * 1e-3, 5) if p_us else None,
               "bound": "hbm (accounting of SURVEY 8(d)); what binds is latency: " +
                        ("one block, a chain of dependent scalar steps" if name == "k_solve" else "dependent cache round trips + instruction issue")}
        if p_us:
            rel = abs(live_ms - p_us * 1e-3) / (p_us * 1e-3)
            ent["live_vs_profiled_rel_diff"] = round(rel, 4)
            if rel > 0.15:
                disagree.append((name, live_ms, p_us * 1e-3))
        kernels.append(ent)
    chain_ms = sum(k_["avg_launch_ms"] for k_ in kernels)
    for k_ in kernels:
        k_["share_of_chain_time"] = round(k_["avg_launch_ms"] / chain_ms, 4) if chain_ms > 0 else None
    for name, live_ms, p_ms in disagree:
        print(f"bench.py: ROOFLINE DISAGREEMENT {name}: measured in this run {live_ms * 1e3:.2f} us, committed rocprofv3 average "
              f"{p_ms * 1e3:.2f} us (> 15 %): the committed profile ({prof.get('_file')}) does not describe this tree / box",
              file=sys.stderr)
    km = next(k_ for k_ in kernels if k_["kernel"] == "k_match2")
    bytes_per_launch = alg["k_match2"]
    dur_ms = km["avg_launch_ms"]
    achieved = km["achieved"]
    # first / converged split of the matcher: bytes and time of the SAME launches on both sides of every fraction
    first_ms = max(first_ev_ms - gap_ms, 1e-6)
    conv_ms = max((dur_ms * iters - first_ms) / max(iters - 1, 1), 1e-6)
    bytes_first, bytes_conv = N * (236.0 + 12.0 * cbar_first), N * (236.0 + 12.0 * cbar_conv)

    def side(nbytes, ms, c_):
        return {"cbar_candidates_per_query": round(c_, 2), "alg_bytes_per_launch": int(nbytes), "avg_launch_ms": round(ms, 5),
                "achieved": round(nbytes / (ms * 1e-3) / 1e9, 2), "frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}

    # measured stream-copy ceiling of this device (SURVEY.md 8(d) asks for it beside the spec peak)
    import ctypes as _C

    from open3d_slam_advanced_rss_2024_public_amd import _lib as _l

    copy_gbs = _C.c_double()
    if _l.lib().o3s_stream_copy_gbs(device, 1 << 31, 5, _C.byref(copy_gbs)) != 0:
        copy_gbs = _C.c_double(0.0)
    # SURVEY 8(d)'s whole-iteration figure: N * (280 + 12 c-bar) bytes per iteration x measured iterations/s
    iter_bytes = N * (280.0 + 12.0 * cbar)
    iter_gbs = iter_bytes * it_per_s_one_pair / 1e9
    roofline = {
        "bound": "hbm", "kernel": "k_match2", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
        "traffic_over_algorithmic": round(traffic / bytes_per_launch, 4) if traffic else None,
        "binding_limit": "instruction issue + dependent L2 / Infinity-Cache round trips (the working set is cache-resident: "
                         "see traffic_over_algorithmic); the HBM model is the accounting SURVEY 8(d) prescribes, not what binds",
        "measured_stream_copy_GBs": round(copy_gbs.value, 1),
        "frac_of_measured_copy": round(achieved / copy_gbs.value, 5) if copy_gbs.value > 0 else None,
        "alg_bytes_per_launch": int(bytes_per_launch),
        "avg_launch_ms": round(dur_ms, 5),
        "avg_launch_ms_source": "measured in this run: HIP events around every launch of 3 eagerly issued chains "
                                f"({kms['match'][1]} launches, first iterations included) minus the per-launch event overhead g, "
                                "g = (sum of the kernels' event pairs - one iteration of the graph-replayed timed chain) / kernels",
        "event_overhead_ms": round(gap_ms, 5), "timed_chain_ms_per_iteration": round(chain_iter_ms, 5),
        "profiled_avg_ms": km["profiled_avg_ms"], "profiled_source": (prof or {}).get("_file"),
        "live_vs_profiled_ok": not any(n_ == "k_match2" for n_, _, _ in disagree) if km["profiled_avg_ms"] else None,
        "largest_kernel_by_time": max(kernels, key=lambda k_: k_["avg_launch_ms"])["kernel"],
        "split": {"first_iteration": side(bytes_first, first_ms, cbar_first), "converged_iterations": side(bytes_conv, conv_ms, cbar_conv),
                  "note": "first = the one launch of a call without incumbents (the far search does the work; icp.yaml's chain stops after "
                          "3-5 iterations, so it is a third of what that chain runs); converged = (all - first) / (iterations - 1), bytes "
                          "and time alike"},
        "whole_iteration": {"alg_bytes_per_iteration": int(iter_bytes), "achieved_GBs": round(iter_gbs, 2),
                            "frac_of_peak": round(iter_gbs / HBM_PEAK_GBS, 5),
                            "frac_of_measured_copy": round(iter_gbs / copy_gbs.value, 5) if copy_gbs.value > 0 else None,
                            "formula": "N * (280 + 12 * c_bar) bytes x iterations/s (SURVEY.md 8(d))"},
        "cbar_candidates_per_query": round(cbar, 2), "rows_per_query": round(rows, 2),
    }
    return roofline, kernels, disagree


def yaml_chain_record(icp_y, pair, calls):
    """The chain open3d_slam runs (icp.yaml: Differential 0.001 / 0.01 / 3 before Counter 15) on a resident pair, call by call:
    three warm-up calls (eager, captured, replayed), then `calls` timed ones — every call's wall time with what ended its waits and
    how the chain went out (o3s_icp_host_split_ex), so that a stall shows as a max / p99 and not as a shifted mean (timed_region:
    the interpreter's garbage collector is held off, what it does anyway is recorded)."""
    for _ in range(3):
        icp_y.compute_resident(pair.T_init, with_trace=False)
    ms, rec = [], []
    with timed_region() as tr:
        t_all = time.perf_counter()
        for _ in range(calls):
            t0 = time.perf_counter()
            Ty = icp_y.compute_resident(pair.T_init, with_trace=False)
            ms.append(1e3 * (time.perf_counter() - t0))
            rec.append(icp_y.host_split_ex())
        t_all = time.perf_counter() - t_all
    ms = np.array(ms)
    dTy = np.linalg.inv(pair.T_gt) @ Ty.astype(np.float64)
    last = rec[-1]
    return {"chain": "icp.yaml: DifferentialTransformationChecker{0.001, 0.01, 3} then CounterTransformationChecker{15}",
            "iterations": int(icp_y.stats.iterations), "calls": calls,
            "ms_per_registration": round(float(np.median(ms)), 4),
            "ms_per_call": {"min": round(float(ms.min()), 4), "median": round(float(np.median(ms)), 4),
                            "p99": round(float(np.percentile(ms, 99)), 4), "max": round(float(ms.max()), 4),
                            "mean": round(float(ms.mean()), 4), "loop_mean": round(1e3 * t_all / calls, 4)},
            "slowest_calls": [dict(call=int(k_), ms=round(float(ms[k_]), 4), host_issue_us=round(rec[k_]["host_issue_us"], 1),
                                   host_wait_us=round(rec[k_]["host_wait_us"], 1), ended_by_stream_guard=rec[k_]["waits_ended_by_stream_guard"])
                              for k_ in np.argsort(ms)[::-1][:3]],
            "gpu_chain_ms": round(icp_y.stats.gpu_ms, 4), "gpu_prepare_ms": round(last["gpu_prepare_us"] * 1e-3, 4),
            "registrations_per_s": round(calls / t_all, 1),
            "python_gc": tr.record(),
            "waits_ended_by": {"post": sum(r["waits_ended_by_post"] for r in rec), "event": sum(r["waits_ended_by_event"] for r in rec),
                               "stream_guard": sum(r["waits_ended_by_stream_guard"] for r in rec)},
            "issued": {k: sum(1 for r in rec if r["issued"] == k) for k in ("eager", "captured", "replayed")},
            "last_call_split_us": {"host_issue": round(last["host_issue_us"], 1), "host_wait": round(last["host_wait_us"], 1),
                                   "queries": last["queries"], "gpu_prepare": round(last["gpu_prepare_us"], 1)},
            "iterations_per_s": round(icp_y.stats.iterations * calls / t_all, 1),
            "pose_error_vs_ground_truth_m": float(np.linalg.norm(dTy[:3, 3]))}


def cpu_model_name() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_record(args, pair, cfg, device):
    """The oracle (oracle/: a port of the reference's algorithm, the CHECKER — never the thing shipped) timed on this box's host
    cores on a bounded sample of the same workload: `cpu_iters` iterations of the same pair, kd-tree matcher with OpenMP over the
    queries on the job's core share, and a single-thread leg (libpointmatcher's loop is single-threaded apart from libnabo)."""
    from oracle import oracle as orc
    from open3d_slam_advanced_rss_2024_public_amd import ICP

    # the GPU box gives one GPU's job a 16-core share of its host CPU; never oversubscribe beyond the affinity mask
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    cpu_iters = max(1, args.cpu_iters)
    ocfg = orc.OracleConfig(use_differential=False, max_iters=cpu_iters)
    o = orc.OracleIcp(ocfg, threads=cores)
    o.init_reference(pair.map_xyz, pair.map_normals)
    tc = time.perf_counter()
    To = o.compute(pair.scan_xyz, pair.scan_normals, pair.T_init)
    tc = time.perf_counter() - tc
    o1 = orc.OracleIcp(orc.OracleConfig(use_differential=False, max_iters=max(1, cpu_iters // 5)), threads=1)
    o1.init_reference(pair.map_xyz, pair.map_normals)
    t1c = time.perf_counter()
    o1.compute(pair.scan_xyz, pair.scan_normals, pair.T_init)
    t1c = time.perf_counter() - t1c
    # agreement GPU <-> CPU after the same number of iterations
    icp_c = ICP(cfg(max_iters=cpu_iters), device=device)
    icp_c.init_reference(pair.map_xyz, pair.map_normals)
    Tg = icp_c.compute(pair.scan_xyz, pair.scan_normals, pair.T_init)
    dt, ang = orc.pose_error(To, Tg)
    icp_c.close()
    return {
        "value": round(cpu_iters / tc, 3), "unit": "ICP iterations/s", "cores": cores, "cpu_model": cpu_model_name(), "kind": "port",
        "sample": f"same C2 pair, {cpu_iters} iterations, exact kd-tree matcher with OpenMP over queries on {cores} "
                  f"threads (match {o.stats.match_ms / cpu_iters:.1f} ms, outlier {o.stats.outlier_ms / cpu_iters:.1f} ms, "
                  f"minimise {o.stats.minimize_ms / cpu_iters:.1f} ms per iteration); single thread: "
                  f"{max(1, cpu_iters // 5) / t1c:.3f} it/s",
        "single_thread_value": round(max(1, cpu_iters // 5) / t1c, 3),
        "gpu_vs_cpu_pose_delta_m": float(np.linalg.norm(dt)), "gpu_vs_cpu_pose_delta_rad": float(ang),
    }


def _c3_make_pair(i):
    """One of BASELINE config 3's 64 (scan, map patch, T_init) triples (seed base + i): runs in a pool of fresh interpreters."""
    from open3d_slam_advanced_rss_2024_public_amd import synthetic as syn

    return syn.make_scan_pair(100_000, 400_000, 0.1, seed=1000 + i)


def measure_c3(args, device, n_pairs=64):
    """BASELINE config 3 on this ONE GPU: 64 independent scan/submap pairs (100 k-point scan vs a 400 k-point map patch each, the
    loop-closure candidates of PlaceRecognition.cpp:70-71), every pair resident, registered with the icp.yaml chain (stops by
    itself) through o3s_icp_compute_batch — all 64 chains in flight, and 8 at a time (what a rank of the 8-GPU node holds)."""
    import multiprocessing as mp

    from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, compute_batch

    t0 = time.perf_counter()
    procs = max(1, min(12, len(os.sched_getaffinity(0)) - 2))
    with mp.get_context("spawn").Pool(procs) as pool:   # spawn: fresh interpreters, nothing of this process's GPU state is inherited
        pairs = pool.map(_c3_make_pair, range(n_pairs), chunksize=2)
    t_gen = time.perf_counter() - t0
    icps, T_init, T_gt = [], [], []
    for sp in pairs:
        icp = ICP(IcpConfig(), device=device)
        assert icp.init_reference(sp.map_xyz, sp.map_normals)
        icp.set_reading(sp.scan_xyz, sp.scan_normals)
        icps.append(icp)
        T_init.append(sp.T_init)
        T_gt.append(sp.T_gt)
    del pairs
    for _ in range(3):   # eager, captured, replayed
        for lo in range(0, n_pairs, 8):
            compute_batch(icps[lo:lo + 8], T_init[lo:lo + 8])
    res = {}
    poses = None
    for in_flight in (8, n_pairs):
        best = None
        for _rep in range(3):
            with timed_region():
                t0 = time.perf_counter()
                got, iters = [], 0
                for lo in range(0, n_pairs, in_flight):
                    p_, codes, stats = compute_batch(icps[lo:lo + in_flight], T_init[lo:lo + in_flight])
                    assert all(c == 0 for c in codes), codes
                    got += p_
                    iters += sum(s_.iterations for s_ in stats)
                dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, iters)
            poses = got
        res[f"in_flight_{in_flight}"] = {"all_pairs_ms": round(1e3 * best[0], 3), "pairs_per_s": round(n_pairs / best[0], 1),
                                        "icp_iterations_per_s": round(best[1] / best[0], 1), "iterations_total": best[1]}
    # every pair equals its single call (same arithmetic, only overlapped), and registers its scan
    singles_equal = True
    for k in (0, n_pairs // 2, n_pairs - 1):
        singles_equal = singles_equal and bool(np.array_equal(icps[k].compute_resident(T_init[k], with_trace=False), poses[k]))
    errs = []
    for k in range(n_pairs):
        d = np.linalg.inv(np.asarray(T_gt[k], np.float64)) @ poses[k].astype(np.float64)
        errs.append(float(np.linalg.norm(d[:3, 3])))
    for icp in icps:
        icp.close()
    best_all = res[f"in_flight_{n_pairs}"]
    return {"workload": f"C3 on one GPU: {n_pairs} pairs, 100000-pt scan vs 400000-pt map patch each, icp.yaml chain (stops by itself, <= 15 iterations), "
                        "all pairs resident, o3s_icp_compute_batch",
            "value": best_all["icp_iterations_per_s"], "unit": "ICP iterations/s (sum over pairs)", "pairs": n_pairs,
            "all_pairs_ms": best_all["all_pairs_ms"], "pairs_per_s": best_all["pairs_per_s"], **res,
            "timing": "best of 3 passes over the 64 pairs per setting",
            "same_pose_as_single_calls": singles_equal, "pose_error_vs_ground_truth_m_max": round(max(errs), 6),
            "fixture_generation_s": round(t_gen, 1), "fixture_generation_processes": procs}


def measure_c5(args):
    """BASELINE config 5's per-scan loop through the COMPILED driver (tests/cpp/mapper_loop.cpp over cpp/o3s_mapper.hpp, plain g++,
    C-ABI library only): tools/mapper_cpp_bench.py as a child process — it ray-casts the sweeps in a pool of its own, builds the
    driver, runs it twice (warm-up, timed) and prints one JSON line.  (i) 300 sweeps, one submap, sweeps pre-processed by the
    receiving thread: the pipeline rate is the headline, with the oracle's host loops over the first sweeps beside it (the
    "end-to-end Hz vs CPU" of BASELINE.json configs[4]); (ii) the closed loop (20 m submaps, loop-closure refinements between
    resident submaps inline on the mapping thread)."""
    import subprocess

    tool = os.path.join(ROOT, "tools", "mapper_cpp_bench.py")

    def run(env_over, timeout):
        env = dict(os.environ)
        env.update(env_over)
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, tool], capture_output=True, text=True, env=env, timeout=timeout)
        if r.returncode != 0:
            return {"error": (r.stderr or r.stdout)[-600:]}
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        d["wall_s_with_generation"] = round(time.perf_counter() - t0, 1)
        return d

    sweeps = int(os.environ.get("O3S_BENCH_C5_SWEEPS", "300"))
    one = run({"SCANS": str(sweeps), "PREFETCH": "2", "PRELOAD": "1", "CPU_SCANS": "8", "ALSO_REF_PERIOD": "2.0"}, 600)
    loop = run({"SCANS": "320", "STEP": "0.5", "LOOP": "1", "SUBMAP_RADIUS": "20", "PREFETCH": "2", "PRELOAD": "1"}, 600)
    # the same sweeps through THREE host stages — one thread stages the raw sweep (page-locked memory), a second pre-processes it, the
    # mapping thread registers and inserts: what the library sustains when the receiving side is not one thread doing a pageable copy
    three = run({"SCANS": str(sweeps), "PREFETCH": "3", "PRELOAD": "1", "PINNED": "1", "ALSO_REF_PERIOD": "2.0"}, 600)
    out = {"workload": f"C5 per-scan loop, compiled driver: {sweeps} ray-cast sweeps (64 x 2048, ~130 k returns), scan and map voxels 0.1 m, icp.yaml chain, "
                       "ICP reference renewed on every sweep, sweeps pre-processed by the receiving thread"}
    if "error" in one:
        out["error"] = one["error"]
    else:
        cpu = one.get("cpu_host_loop") or {}
        out.update({"value": one["pipeline_hz_steady_state"], "unit": "sweeps/s (pipeline, steady state)",
                    "pipeline_hz_steady_state": one["pipeline_hz_steady_state"], "ms_per_call_median": one["ms_per_scan_median"],
                    "ms_per_call_p90_p99_max": one["ms_per_scan_p90_p99_max"], "calls_per_s": one["hz"],
                    "mapper_stopwatches_ms_median": one["mapper_stopwatches_ms_median"], "producer_ms_median": one["producer_ms_median"],
                    "icp_iterations_median": one["icp_iterations_median"], "pose_error_m_max": one["pose_error_m_max"],
                    "cpu_host_loop": cpu, "gpu_vs_cpu_hz": round(one["pipeline_hz_steady_state"] / cpu["hz"], 1) if cpu.get("hz") else None,
                    # the same sweeps with the renewal period every parameter file of the reference sets (reference_cloud_seting_period = 2.0 s
                    # at 10 sweeps/s: the matcher is re-initialised every 20th sweep, Mapper.cpp:349-366); the headline renews on EVERY sweep
                    "reference_renewed_every_2_s": one.get("also_with_reference_renewal_period"),
                    "wall_s_with_generation": one["wall_s_with_generation"]})
    if "error" in three:
        out["three_stages_page_locked"] = {"error": three["error"]}
    else:
        # `value` of this record = the better of the two drivers' pipeline rates with the reference renewed on EVERY sweep: the two-stage
        # driver's rate hangs on how fast the box's host copies a pageable 6 MB sweep (0.39 - 0.65 ms: 1 570 - 2 010 Hz over the boxes seen),
        # the three-stage driver with page-locked sweeps does not.  Both are in the record; the per-call figures above stay the two-stage ones.
        if "error" not in one and three["pipeline_hz_steady_state"] > out.get("value", 0.0):
            out["two_stages_pageable_pipeline_hz"] = out["value"]
            out["value"] = three["pipeline_hz_steady_state"]
            out["value_is"] = "three host stages, page-locked sweeps, reference renewed on every sweep (three_stages_page_locked)"
            if out.get("cpu_host_loop", {}).get("hz"):
                out["gpu_vs_cpu_hz"] = round(out["value"] / out["cpu_host_loop"]["hz"], 1)
        elif "error" not in one:
            out["value_is"] = "two host stages, pageable sweeps, reference renewed on every sweep"
        t2 = three.get("also_with_reference_renewal_period") or {}
        out["three_stages_page_locked"] = {
            "workload": "the same sweeps, three host stages: a thread stages sweep k + 2 from page-locked memory (o3s_raw_scan_upload), a second pre-processes sweep k + 1 "
                        "(o3s_scan_preprocess_staged), the mapping thread registers and inserts sweep k",
            "pipeline_hz_steady_state": three["pipeline_hz_steady_state"], "ms_per_call_median": three["ms_per_scan_median"],
            "ms_per_call_p90_p99_max": three["ms_per_scan_p90_p99_max"], "stage_ms_median": {"staging": three["producer_ms_median"], "pre-processing": three.get("second_stage_ms_median")},
            "pose_error_m_max": three["pose_error_m_max"],
            "reference_renewed_every_2_s": {k_: t2.get(k_) for k_ in ("pipeline_hz_steady_state", "ms_per_scan_median", "ms_per_scan_p90_p99_max", "pose_error_m_max")}}
    if "error" in loop:
        out["closed_loop"] = {"error": loop["error"]}
    else:
        cl = loop.get("loop_closures") or []
        out["closed_loop"] = {"workload": "320 sweeps 0.5 m apart around the block (160 m, one lap and a bit), 20 m submaps, loop-closure refinements inline",
                              "submaps": loop["submaps"], "pipeline_hz_steady_state": loop["pipeline_hz_steady_state"],
                              "ms_per_call_median": loop["ms_per_scan_median"], "ms_per_call_p90_p99_max": loop["ms_per_scan_p90_p99_max"],
                              "refinements": len(cl), "refinement_ms": [round(c_["ms"], 3) for c_ in cl],
                              "refinement_overlap_points": [c_["overlap_points"] for c_ in cl], "refinement_updates": [c_["updates"] for c_ in cl],
                              "pose_error_m_max": loop["pose_error_m_max"], "wall_s_with_generation": loop["wall_s_with_generation"]}
    return out


def measure_c4(args, device):
    """BASELINE config 4 (500k-pt scan vs 20M-pt map, 0.02 m voxels, 50 iterations) timed in the same run: the configuration
    in which the HBM roofline is the right ruler.  Same procedure as the headline (warm-up, timed steps of the graph-replayed
    chain, pose check), its own roofline object with the first / converged split."""
    from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig
    from open3d_slam_advanced_rss_2024_public_amd import synthetic as syn

    N, M, voxel, iters = 500_000, 20_000_000, 0.02, 50
    t0 = time.time()
    pair = syn.make_scan_pair(N, M, voxel, seed=0)
    t_gen = time.time() - t0

    def cfg(**kw):
        base = dict(use_differential=False, max_iters=iters)
        base.update(kw)
        return IcpConfig(**base)

    icp = ICP(cfg(), device=device)
    t0 = time.time()
    assert icp.init_reference(pair.map_xyz, pair.map_normals)
    t_init = time.time() - t0
    icp.set_reading(pair.scan_xyz, pair.scan_normals)
    for _ in range(2):
        icp.compute_resident(pair.T_init, with_trace=False)
    steps = 6
    with timed_region():
        t0 = time.perf_counter()
        for _ in range(steps):
            T = icp.compute_resident(pair.T_init, with_trace=False)
        elapsed = time.perf_counter() - t0
    value = iters * steps / elapsed
    gpu_ms = icp.stats.gpu_ms
    dT = np.linalg.inv(pair.T_gt) @ T.astype(np.float64)
    roofline, kernels, _ = roofline_of(icp, cfg, pair, N, M, voxel, iters, value, gpu_ms, device)
    # the chain icp.yaml runs, on this pair
    icp_y = ICP(IcpConfig(), device=device)
    icp_y.init_reference(pair.map_xyz, pair.map_normals)
    icp_y.set_reading(pair.scan_xyz, pair.scan_normals)
    yaml_c4 = yaml_chain_record(icp_y, pair, calls=10)
    out = {"workload": f"C4: {N}-pt scan vs {M}-pt voxel map, {voxel} m voxels, {iters} iters, icp.yaml chain (Trimmed 0.9)",
           "value": round(value, 2), "unit": "ICP iterations/s", "steps": steps, "ms_per_step": round(1e3 * elapsed / steps, 4),
           "gpu_chain_ms_per_step": round(gpu_ms, 4), "correspondences_per_s": round(value * N, 1),
           "pose_error_vs_ground_truth_m": float(np.linalg.norm(dT[:3, 3])), "kept_pairs": int(icp.stats.kept_pairs),
           "roofline": roofline, "roofline_kernels": kernels,
           "icp_yaml_chain": yaml_c4,
           "init_reference_s": round(t_init, 3), "fixture_generation_s": round(t_gen, 2)}
    icp_y.close()
    icp.close()
    return out


def measure_sharded_extra(args, rank, world, device, dist, torch):
    """N > 1: SURVEY 8(e) mode 2 on the same ranks, after the pairs measurement: ONE C2 pair, the reading split over the ranks,
    the three exchanges of every iteration as ncclAllReduce calls issued from C on the kernel stream (libo3dslam_icp_rccl.so).
    Reports iterations/s of that single registration (strong scaling), the collectives RCCL saw and the bytes they moved."""
    import ctypes as C

    from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, _lib
    from open3d_slam_advanced_rss_2024_public_amd import synthetic as syn
    from open3d_slam_advanced_rss_2024_public_amd.parallel import shard_slice

    N, M, iters = args.scan, args.map, args.iters
    pair = syn.make_scan_pair(N, M, args.voxel, seed=0)  # the same pair on every rank
    R = _lib.rccl_lib()
    uid = C.create_string_buffer(128)
    if rank == 0:
        assert R.o3s_rccl_unique_id(uid) == 0, R.o3s_rccl_last_error()
    t = torch.frombuffer(bytearray(uid.raw), dtype=torch.uint8).to(f"cuda:{device}")
    dist.broadcast(t, src=0)
    uid = C.create_string_buffer(bytes(t.cpu().numpy().tobytes()), 128)
    comm = C.c_void_p()
    assert R.o3s_rccl_create(uid, rank, world, device, C.byref(comm)) == 0, R.o3s_rccl_last_error()
    icp = ICP(IcpConfig(use_differential=False, max_iters=iters), device=device)
    assert icp.init_reference(pair.map_xyz, pair.map_normals)
    sl = shard_slice(N, world, rank)
    icp.set_reading(pair.scan_xyz[sl], pair.scan_normals[sl])
    icp.shard_configure_rccl(N, rank, world, comm.value)
    for _ in range(3):
        icp.compute_resident(pair.T_init, with_trace=False)
    dist.barrier()
    torch.cuda.synchronize()
    before = int(R.o3s_rccl_collectives(comm))
    steps = max(3, min(10, args.steps))
    with timed_region():
        t0 = time.perf_counter()
        for _ in range(steps):
            T = icp.compute_resident(pair.T_init, with_trace=False)
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
    tt = torch.tensor([t1 - t0], dtype=torch.float64, device=f"cuda:{device}")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    issued = int(R.o3s_rccl_collectives(comm)) - before   # 0 once the chain replays from a hipGraph: the collectives are graph nodes

    def make_sharded(kw):
        c = IcpConfig(use_differential=False, max_iters=iters)
        for k_, v_ in kw.items():
            setattr(c, k_, v_)
        h_ = ICP(c, device=device)
        assert h_.init_reference(pair.map_xyz, pair.map_normals)
        h_.set_reading(pair.scan_xyz[sl], pair.scan_normals[sl])
        h_.shard_configure_rccl(N, rank, world, comm.value)
        return h_

    roofline = sharded_roofline(icp, make_sharded, pair, sl.stop - sl.start, iters)
    L = _lib.lib()
    L.o3s_icp_shard_bytes_per_iteration.restype = C.c_int64
    L.o3s_icp_shard_bytes_per_iteration.argtypes = [C.c_int32, C.c_int64]
    out = None
    if rank == 0:
        dT = np.linalg.inv(pair.T_gt) @ T.astype(np.float64)
        out = {"mode": "ONE pair sharded (SURVEY 8(e) mode 2): reading split over the ranks, reference replicated",
               "value": round(iters * steps / elapsed, 2), "unit": "ICP iterations/s (one registration, strong scaling)",
               "ranks": world, "rccl_ranks": world, "ms_per_step": round(1e3 * elapsed / steps, 4), "collectives_per_iteration": 3,
               "roofline": roofline,
               "bytes_per_iteration_per_rank": int(L.o3s_icp_shard_bytes_per_iteration(world, N)),
               "rccl_collectives_issued_from_host_during_timed_steps": issued,
               "rccl_collectives_total": int(R.o3s_rccl_collectives(comm)),
               "exchange": "ncclAllReduce from C on the kernel stream (libo3dslam_icp_rccl.so), captured into the chain's hipGraph",
               "pose_error_vs_ground_truth_m": float(np.linalg.norm(dT[:3, 3]))}
    icp.close()
    R.o3s_rccl_destroy(comm)
    return out


def _gpu_count() -> int:
    """GPUs of this node WITHOUT initialising the HIP runtime in this process (the ranks are started as children)."""
    import glob

    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(f) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0 and int(props.get("gfx_target_version", "0")) > 0:
                n += 1
        except OSError:
            pass
    return n


def _self_launch(args) -> int:
    """`python3 bench.py --gpus N` (N > 1) outside torch.distributed.run: start the N ranks as children of this process
    (one per GPU, RCCL rendezvous on 127.0.0.1), relay rank 0's single JSON line, exit with the children's status.
    Nothing in this process touches the GPU, so no initialised runtime is ever replaced by another program."""
    import socket
    import subprocess

    have = _gpu_count()
    if have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but this node exposes {have} GPU(s)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if proc.returncode != 0 or not lines:
        print(f"bench.py: the {args.gpus}-rank run failed (exit {proc.returncode})", file=sys.stderr)
        return proc.returncode or 1
    print(lines[-1], flush=True)
    return 0


def main():
    if "WORLD_SIZE" not in os.environ:
        args = parse()
        if args.gpus > 1:
            sys.exit(_self_launch(args))
    # stdout carries exactly ONE line (the JSON record): everything native libraries print while the run is in flight —
    # RCCL writes a five-line version banner to stdout when the first communicator is created — goes to stderr instead.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        line = _run()
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if line is not None:
        print(line, flush=True)


def _metric_name(scan: int, map_: int) -> str:
    """BASELINE.json's metric on its configuration; other sizes (e.g. C4) say what they are."""
    def short(n):
        return f"{n // 1_000_000}M" if n % 1_000_000 == 0 else f"{n // 1000}k" if n % 1000 == 0 else str(n)
    return f"ICP iterations/s ({short(scan)}-pt scan vs {short(map_)}-pt map)"


def _run():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1):
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    import torch

    dist = None
    if world > 1 or os.environ.get("O3S_BENCH_FORCE_DIST") == "1":  # the env knob rehearses the N > 1 code path on one GPU
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    device = local_rank if world > 1 else 0
    torch.cuda.set_device(device)

    from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig
    from open3d_slam_advanced_rss_2024_public_amd import synthetic as syn

    N, M, iters = args.scan, args.map, args.iters
    if args.mode == "sharded":
        return run_sharded(args, rank, world, device, dist, torch)
    t_gen = time.time()
    pair = syn.make_scan_pair(N, M, args.voxel, seed=rank)
    t_gen = time.time() - t_gen

    def cfg(**kw):
        base = dict(use_differential=False, max_iters=iters, grid_cell=args.grid_cell, sort_queries=not args.no_sort,
                    use_graph=not args.no_graph)
        base.update(kw)
        return IcpConfig(**base)

    icp = ICP(cfg(), device=device)
    t0 = time.time()
    assert icp.init_reference(pair.map_xyz, pair.map_normals)
    t_init_ref = time.time() - t0
    icp.set_reading(pair.scan_xyz, pair.scan_normals)
    # --pairs-per-gpu P > 1: P independent pairs per rank (BASELINE config 3: 64 pairs = 8 per GPU on the 8-GPU node), all chains
    # in flight at once through o3s_icp_compute_batch (one stream per pair); still no collective in the data path
    P = max(1, args.pairs_per_gpu)
    extra_handles, T_inits = [], [pair.T_init]
    if P > 1:
        from open3d_slam_advanced_rss_2024_public_amd import compute_batch

        for k in range(1, P):
            pk = syn.make_scan_pair(N, M, args.voxel, seed=1000 * (rank + 1) + k)
            hk = ICP(cfg(), device=device)
            assert hk.init_reference(pk.map_xyz, pk.map_normals)
            hk.set_reading(pk.scan_xyz, pk.scan_normals)
            extra_handles.append(hk)
            T_inits.append(pk.T_init)
            del pk
        all_handles = [icp] + extra_handles

        def step():
            poses, codes, _ = compute_batch(all_handles, T_inits)
            assert all(c == 0 for c in codes), codes
            return poses[0]
    else:
        def step():
            return icp.compute_resident(pair.T_init, with_trace=False)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    with timed_region() as tr_main:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            T = step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
    gpu_ms_chain = icp.stats.gpu_ms
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{device}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_iters = iters * args.steps * world * P
    value = total_iters / elapsed
    for hk in extra_handles:
        hk.close()

    sharded_extra = None
    if dist is not None and not args.timing_only and not args.no_sharded_extra and (N, M) == (100_000, 2_000_000):
        # the one driver run on N GPUs measures BOTH modes of SURVEY 8(e): every rank takes part (collective calls inside)
        try:
            sharded_extra = measure_sharded_extra(args, rank, world, device, dist, torch)
        except Exception as e:  # noqa: BLE001 — the pairs line must survive a failure of the extra
            sharded_extra = {"error": f"{type(e).__name__}: {e}"}
    out = None
    if rank == 0 and args.timing_only:
        out = {"metric": _metric_name(args.scan, args.map), "value": round(value, 2), "n_gpus": world,
               "ms_per_step": round(1e3 * elapsed / args.steps, 4), "timing_only": True}
    elif rank == 0:
        # pose sanity: the benchmarked run must actually register the scan
        dT = np.linalg.inv(pair.T_gt) @ T.astype(np.float64)
        pose_err_m = float(np.linalg.norm(dT[:3, 3]))

        roofline, kernels, disagree = roofline_of(icp, cfg, pair, N, M, args.voxel, iters, value / (world * P), gpu_ms_chain, device)
        # ---- the chain open3d_slam actually runs (icp.yaml: Differential 0.001 / 0.01 / 3 before Counter 15), same pair ----
        yaml_chain = None
        if args.iters == 50:
            icp_y = ICP(IcpConfig(grid_cell=args.grid_cell, sort_queries=not args.no_sort, use_graph=not args.no_graph), device=device)
            icp_y.init_reference(pair.map_xyz, pair.map_normals)
            icp_y.set_reading(pair.scan_xyz, pair.scan_normals)
            yaml_chain = yaml_chain_record(icp_y, pair, calls=40)
            icp_y.close()

        # ---- PCIe-inclusive rate (host buffers handed over every call); never the headline value ----
        for _ in range(3):  # the first calls allocate the upload buffer and re-capture the chain's graph behind it
            icp.compute(pair.scan_xyz, pair.scan_normals, pair.T_init)
        reps = max(2, min(5, args.steps))
        with timed_region():
            t1 = time.perf_counter()
            for _ in range(reps):
                icp.compute(pair.scan_xyz, pair.scan_normals, pair.T_init)
            pcie_value = iters * reps / (time.perf_counter() - t1)

        # ---- several pairs in flight on ONE GPU (BASELINE config 3's per-GPU share: o3s_icp_compute_batch, one stream per
        # pair).  Reported beside the headline, never as `value`: the chains of different pairs overlap, so the GPU's
        # idle slots between one pair's dependent kernels get filled. ----
        batched = None
        if args.batch_pairs > 1:
            from open3d_slam_advanced_rss_2024_public_amd import compute_batch
            PB = args.batch_pairs   # NOT `P`: that is the timed run's pairs per GPU and goes into `config`
            handles = [ICP(cfg(), device=device) for _ in range(PB)]
            for hnd in handles:
                hnd.init_reference(pair.map_xyz, pair.map_normals)
                hnd.set_reading(pair.scan_xyz, pair.scan_normals)
            Tin = [pair.T_init] * PB
            for _ in range(3):
                compute_batch(handles, Tin)
            breps = max(3, min(10, args.steps))
            with timed_region():
                tb = time.perf_counter()
                for _ in range(breps):
                    poses, codes, _st = compute_batch(handles, Tin)
                tb = time.perf_counter() - tb
            batched = {"pairs_in_flight": PB, "value": round(PB * iters * breps / tb, 1), "unit": "ICP iterations/s (sum over pairs)",
                       "all_ok": bool(all(c == 0 for c in codes)),
                       "same_pose_as_single": bool(all(np.array_equal(p_, T) for p_ in poses))}
            for hnd in handles:
                hnd.close()

        # ---- CPU baseline: the oracle (a port of the reference's algorithm) on this box's host cores ----
        cpu = None
        if not args.no_cpu and world == 1:  # the CPU leg belongs to the N = 1 line only
            cpu = cpu_baseline_record(args, pair, cfg, device)
        c4 = None
        if world == 1 and dist is None and not args.no_c4 and (N, M) == (100_000, 2_000_000) and args.voxel == 0.1:
            c4 = measure_c4(args, device)
        c3 = c5 = None
        if world == 1 and dist is None and (N, M) == (100_000, 2_000_000) and args.voxel == 0.1:
            for name, on, fn in (("c3", not args.no_c3, lambda: measure_c3(args, device)), ("c5", not args.no_c5, lambda: measure_c5(args))):
                if not on:
                    continue
                try:
                    val = fn()
                except Exception as e:  # noqa: BLE001 — the headline line must survive a failure of an extra
                    val = {"error": f"{type(e).__name__}: {e}"}
                if name == "c3":
                    c3 = val
                else:
                    c5 = val
        from open3d_slam_advanced_rss_2024_public_amd import _lib as _l2
        out = {
            "metric": _metric_name(args.scan, args.map), "value": round(value, 2), "unit": "ICP iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{'C2' if (N, M) == (100_000, 2_000_000) else 'C4' if (N, M) == (500_000, 20_000_000) else 'custom'}: {N}-pt scan vs {M}-pt voxel map, {args.voxel} m voxels, {iters} iters, icp.yaml chain "
                                   "(KDTree maxDist 0.5 exact, Trimmed 0.9, SurfaceNormal 1.57, PointToPlane)",
                       "scan_points": N, "map_points": M, "iterations_per_step": iters, "pairs_per_gpu": P,
                       "parallelism": f"{world * P} independent scan/map pair{'s' if world * P > 1 else ''}, {P} per GPU"
                                      + (" in flight at once (o3s_icp_compute_batch)" if P > 1 else "") + ", no data-path collective"},
            "correspondences_per_s": round(value * N, 1),
            "roofline": roofline,
            "roofline_kernels": kernels,
            "cpu_baseline": cpu,
            "extra": {"pcie_inclusive_value": round(pcie_value, 2), "gpu_chain_ms_per_step": round(gpu_ms_chain, 4),
                      "init_reference_s": round(t_init_ref, 3), "fixture_generation_s": round(t_gen, 2),
                      "pose_error_vs_ground_truth_m": pose_err_m, "kept_pairs": int(icp.stats.kept_pairs),
                      "python_gc_in_timed_steps": tr_main.record(),
                      "batched_on_one_gpu": batched, "icp_yaml_chain": yaml_chain, "c3": c3, "c4": c4, "c5": c5, "sharded_one_pair": sharded_extra,
                      "rocm_runtime": _l2.loaded_rocm_runtimes()},
        }
        strict_fail = bool(args.strict and disagree)
    icp.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and locals().get("strict_fail"):
        print(json.dumps(out), flush=True, file=sys.stderr)
        sys.exit(3)
    return json.dumps(out) if rank == 0 else None


if __name__ == "__main__":
    main()
