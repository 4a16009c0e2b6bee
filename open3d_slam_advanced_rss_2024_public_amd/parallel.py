"""Multi-GPU execution of the ICP path: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the
MI355X node, "gloo" in CPU tests).

The path shards by INDEPENDENT UNITS — scan/submap pairs (BASELINE config 3; loop-closure candidates are a serial loop
in the reference, open3d_slam/src/PlaceRecognition.cpp:70-71).  Pairs are dealt round-robin to ranks, every rank runs
its pairs with no data-path collective (o3s_icp_compute_batch overlaps them on its GPU), and one fixed-size all_gather
at the end returns every pose to every rank.  Nothing here computes ICP: a ``runner`` callable does (the GPU runner
below, or — in CPU tests only — the oracle).
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import numpy as np


def shard_indices(n_units: int, world: int, rank: int) -> List[int]:
    """Round-robin ownership: unit u belongs to rank u % world."""
    return list(range(rank, n_units, world))


def gpu_runner(config, device: int) -> Callable:
    """Returns runner(pairs) -> [(pose 4x4 | None, status, iterations)] that runs the pairs concurrently on `device`."""
    from .icp import ICP, compute_batch

    def run(pairs):
        icps = []
        for p in pairs:
            icp = ICP(config, device=device)
            if not icp.init_reference(p["map_xyz"], p["map_normals"]):
                raise RuntimeError("empty reference")
            icp.set_reading(p["scan_xyz"], p["scan_normals"])
            icps.append(icp)
        poses, codes, stats = compute_batch(icps, [p["T_init"] for p in pairs])
        out = [(poses[k], codes[k], stats[k].iterations) for k in range(len(pairs))]
        for icp in icps:
            icp.close()
        return out

    return run


def run_pairs_sharded(pairs: Sequence[dict], runner: Callable, dist=None, device=None):
    """Every rank passes the SAME list of pairs; returns, on every rank, the results of all pairs in list order.

    ``dist``: an initialised torch.distributed module (None = single process).  The only collective is the final
    all_gather of [n_max, 18] fp32 rows (16 pose floats, status, iterations) per rank."""
    n = len(pairs)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return runner(list(pairs))
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    mine = shard_indices(n, world, rank)
    local = runner([pairs[u] for u in mine]) if mine else []
    n_max = (n + world - 1) // world
    buf = np.zeros((n_max, 18), np.float32)
    buf[:, 16] = -1.0  # padding marker
    for k, (T, status, iters) in enumerate(local):
        if T is not None:
            buf[k, :16] = np.asarray(T, np.float32).reshape(16)
        buf[k, 16] = float(status)
        buf[k, 17] = float(iters)
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    t = torch.from_numpy(buf).to(dev)
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    out = [None] * n
    for r in range(world):
        g = gathered[r].cpu().numpy()
        for k, u in enumerate(shard_indices(n, world, r)):
            status = int(g[k, 16])
            T = g[k, :16].reshape(4, 4).copy() if status == 0 else None
            out[u] = (T, status, int(g[k, 17]))
    return out
