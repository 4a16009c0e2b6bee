"""Multi-GPU execution of the ICP path: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the
MI355X node, "gloo" in CPU tests).

The path shards by INDEPENDENT UNITS — scan/submap pairs (BASELINE config 3; loop-closure candidates are a serial loop
in the reference, open3d_slam/src/PlaceRecognition.cpp:70-71).  Pairs are dealt round-robin to ranks, every rank runs
its pairs with no data-path collective (o3s_icp_compute_batch overlaps them on its GPU), and one fixed-size all_gather
at the end returns every pose to every rank.  Nothing here computes ICP: a ``runner`` callable does (the GPU runner
below, or — in CPU tests only — the oracle).

The second mode — ONE pair sharded over the ranks (``PairSharded``) — splits the reading, replicates the reference and
all-reduces three small quantities per iteration (histograms of the trim selection, kept-pair sums, the 27 sums of the
normal equations) through torch.distributed on the library's own exchange buffer; see include/o3s_icp.h.
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


def shard_slice(n: int, world: int, rank: int) -> slice:
    """Contiguous balanced slice of a reading of n points owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return slice(lo, lo + base + (1 if rank < rem else 0))


class PairSharded:
    """One scan/map pair registered by all ranks of a process group together (SURVEY.md 8(e) mode 2).

    Every rank passes the SAME map, scan and initial guess; rank r keeps slice r of the scan resident on its GPU next
    to a full copy of the map index.  The exchange buffer is a torch tensor, so the collectives are plain
    ``dist.all_reduce`` calls: backend "nccl" (= RCCL over xGMI) reduces it in place on the device, ordered on the
    stream the kernels run on; backend "gloo" (tests: several ranks sharing one GPU) stages it through host memory."""

    def __init__(self, config, device: int, group=None):
        import torch
        import torch.distributed as dist

        from . import _lib
        from .icp import ICP

        self._torch, self._dist, self._group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.dev = torch.device("cuda", device)
        self.icp = ICP(config, device=device)
        self._stream = torch.cuda.Stream(self.dev)  # a real (non-null) stream shared by the kernels and the collectives
        self.icp.set_stream(self._stream.cuda_stream)
        nbytes = int(_lib.lib().o3s_icp_shard_exchange_bytes())
        self._xbuf = torch.zeros(nbytes, dtype=torch.uint8, device=self.dev)
        self._xi32 = self._xbuf.view(torch.int32)
        self._xf64 = self._xbuf.view(torch.float64)
        self._on_device = dist.get_backend(group) == "nccl"
        self.n_total = 0
        self.collectives = 0

    def _allreduce(self, off, count, dtype, _ptr, _stream):
        torch, dist = self._torch, self._dist
        t = self._xi32[off // 4: off // 4 + count] if dtype == 0 else self._xf64[off // 8: off // 8 + count]
        with torch.cuda.stream(self._stream):
            if self._on_device:
                dist.all_reduce(t, group=self._group)
            else:
                c = t.cpu()  # waits for the kernels enqueued so far on this stream
                dist.all_reduce(c, group=self._group)
                t.copy_(c)
        self.collectives += 1

    def init_reference(self, map_xyz, map_normals) -> bool:
        return self.icp.init_reference(map_xyz, map_normals)

    def set_reading(self, scan_xyz, scan_normals):
        n = int(np.asarray(scan_xyz).shape[0])
        if n < self.world:
            raise ValueError("the reading must hold at least one point per rank")
        sl = shard_slice(n, self.world, self.rank)
        self.n_total = n
        self.icp.set_reading(np.asarray(scan_xyz)[sl], None if scan_normals is None else np.asarray(scan_normals)[sl])
        self.icp.shard_configure(n, self.rank, self.world, self._allreduce, self._xbuf.data_ptr())

    def compute(self, T_init, with_trace: bool = True):
        """Same pose, iteration count and statistics on every rank."""
        T = self.icp.compute_resident(T_init, with_trace=with_trace)
        return T

    @property
    def stats(self):
        return self.icp.stats

    def close(self):
        self.icp.close()
