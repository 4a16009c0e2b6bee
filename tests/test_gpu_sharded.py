"""One scan/map pair sharded over several ranks (SURVEY.md 8(e) mode 2; include/o3s_icp.h o3s_icp_shard_configure).
MI355X only.  The one-GPU box validates the exchange protocol with several processes SHARING cuda:0 over gloo (the
collective payloads are staged through host memory) and with RCCL at world size 1; the 8-GPU run uses the same
kernels and the same call sequence with backend "nccl".

Bars: every rank returns the bit-identical pose / iteration count; trim limits and kept-pair counts of every iteration
equal the unsharded chain's (integer work, exact); pose within 1e-6 of the unsharded chain (the fp64 sums are formed
in a different order before their single rounding to fp32) and within 1e-4 m / 1e-4 rad of the CPU oracle."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig, synthetic as syn

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_world(world, backend="gloo", case="yaml"):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, REPO_ROOT=ROOT, OMP_NUM_THREADS="2", SHARD_BACKEND=backend, SHARD_CASE=case,
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_sharded_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def check_against_single(r):
    assert r["same_on_all_ranks"] and r["error"] is None
    assert r["iterations"] == r["iters_single"] == r["iters_oracle"]
    assert r["limits_equal"] and r["kept_equal"]
    assert r["kept"] == r["kept_single"] and r["matched"] == r["matched_single"]
    assert r["ratio"] == r["ratio_single"]
    # the sharded chain forms exact fp64 products of raw moments and centres them once; the unsharded chain (like the reference) rounds
    # p - mean and every product per pair in fp32 — its own rounding noise is what separates the two: <= 1e-6 m on well-conditioned
    # pairs (C1 / C2-like), 9e-6 m after ten iterations on the dense 2 cm case; both sit within 1e-4 of the oracle (next line)
    assert r["dt_single"] <= 2e-5 and r["ang_single"] <= 2e-5
    assert r["dt_oracle"] <= 1e-4 and r["ang_oracle"] <= 1e-4


def test_world1_noop_exchange_equals_unsharded_chain():
    """world = 1 with an identity exchange: the sharded kernel sequence alone."""
    sp = syn.make_scan_pair(5000, 40000, 0.1, seed=5)
    cfg = IcpConfig()
    a = ICP(cfg)
    b = ICP(cfg)
    for h in (a, b):
        assert h.init_reference(sp.map_xyz, sp.map_normals)
        h.set_reading(sp.scan_xyz, sp.scan_normals)
    calls = []
    b.shard_configure(sp.scan_xyz.shape[0], 0, 1, lambda off, count, dtype, ptr, stream: calls.append((off, count, dtype)))
    Ta = a.compute_resident(sp.T_init)
    Tb = b.compute_resident(sp.T_init)
    assert a.stats.iterations == b.stats.iterations
    # three exchanges per iteration (round 5; rounds 3-4: four), each a region of the exchange buffer reduced in place
    # (csrc/icp_shard_kernels.h): the level-1 replicas (R = 16 >> floor(log2(world)) of them: 16 at world size 1, 2 at eight ranks), the
    # level-2 histogram (13 bits in this mode), and region M: level-3 counts, the per-bin raw moments and the block partials of the
    # raw moments of the pairs kept whatever the limit's last seven bits are
    assert len(calls) % 3 == 0 and len(calls) >= 3 * b.stats.iterations
    nb_part = (calls[2][1] - 128 - 34 * 128) // 34
    m_bytes = (128 + 34 * 128 + 34 * 512) * 8
    i_off = m_bytes
    assert calls[:3] == [(i_off, 16 * 2048, 0), (i_off + 16 * 2048 * 4, 8192, 0), (0, 128 + 34 * 128 + 34 * nb_part, 1)]
    assert 1 <= nb_part <= 512
    import ctypes as C
    from open3d_slam_advanced_rss_2024_public_amd import _lib
    f = _lib.lib().o3s_icp_shard_bytes_per_iteration
    f.restype, f.argtypes = C.c_int64, [C.c_int32, C.c_int64]
    n = sp.scan_xyz.shape[0]
    assert f(1, n) == sum(c * (4 if d == 0 else 8) for _, c, d in calls[:3])
    # at eight ranks the replicas that travel are 2, the block partials those of an eighth of the reading: 16.4 + 32.8 + 42.6 KB for C2
    assert f(8, 100_000) == 2 * 2048 * 4 + 8192 * 4 + (128 + 34 * 128 + 34 * 25) * 8 == 91792 and f(8, 100_000) < 312 * 1024 // 3
    # same first limit (same pose: the same element), same kept counts; later limits follow poses that differ in their last bits —
    # the sharded chain centres its raw moments algebraically in fp64, the unsharded one every pair in fp32 before it multiplies
    assert a.stats.trace_limit[0] == b.stats.trace_limit[0] and np.allclose(a.stats.trace_limit, b.stats.trace_limit, rtol=2e-5, atol=0.0)
    assert np.array_equal(a.stats.trace_kept, b.stats.trace_kept)
    assert np.abs(Ta - Tb).max() <= 1e-6
    b.shard_disable()
    Tc = b.compute_resident(sp.T_init)
    assert np.array_equal(Ta, Tc)


def test_exchange_failure_is_reported():
    sp = syn.make_scan_pair(2000, 20000, 0.1, seed=6)
    b = ICP(IcpConfig())
    assert b.init_reference(sp.map_xyz, sp.map_normals)
    b.set_reading(sp.scan_xyz, sp.scan_normals)

    def boom(*_a):
        raise RuntimeError("link down")

    b.shard_configure(sp.scan_xyz.shape[0], 0, 1, boom)
    with pytest.raises(RuntimeError):
        b.compute_resident(sp.T_init)


@pytest.mark.parametrize("world,case", [(2, "yaml"), (3, "fixed"), (2, "notrim"), (2, "c4"), (2, "uneven")])
def test_gloo_ranks_sharing_one_gpu(world, case):
    r = run_world(world, "gloo", case)
    assert r["world"] == world
    check_against_single(r)
    per_iter = 2 if case == "notrim" else 3
    assert r["collectives"] % per_iter == 0 and r["collectives"] >= per_iter * r["iterations"]


def test_all_ranks_fail_together_when_nothing_matches():
    r = run_world(2, "gloo", "far")
    assert r["same_on_all_ranks"] and r["error"] == "ConvergenceError" and r["error_single"] == "ConvergenceError"


def test_rccl_world1():
    """RCCL in the loop: backend "nccl" on the one GPU of this box (in-place ncclAllReduce on the kernel stream)."""
    check_against_single(run_world(1, "nccl", "yaml"))


def test_native_rccl_exchange_world1():
    """libo3dslam_icp_rccl.so: the collectives are ncclAllReduce calls issued from C on the kernel stream."""
    import ctypes as C

    from open3d_slam_advanced_rss_2024_public_amd import _lib

    R = _lib.rccl_lib()
    uid = C.create_string_buffer(128)
    assert R.o3s_rccl_unique_id(uid) == 0, R.o3s_rccl_last_error()
    comm = C.c_void_p()
    assert R.o3s_rccl_create(uid, 0, 1, 0, C.byref(comm)) == 0, R.o3s_rccl_last_error()
    sp = syn.make_scan_pair(5000, 40000, 0.1, seed=5)
    a, b = ICP(IcpConfig()), ICP(IcpConfig())
    for h in (a, b):
        assert h.init_reference(sp.map_xyz, sp.map_normals)
        h.set_reading(sp.scan_xyz, sp.scan_normals)
    b.shard_configure_rccl(sp.scan_xyz.shape[0], 0, 1, comm.value)
    Ta = a.compute_resident(sp.T_init)
    Tb = b.compute_resident(sp.T_init)
    assert a.stats.iterations == b.stats.iterations
    assert R.o3s_rccl_collectives(comm) >= 3 * b.stats.iterations
    assert a.stats.trace_limit[0] == b.stats.trace_limit[0] and np.allclose(a.stats.trace_limit, b.stats.trace_limit, rtol=2e-5, atol=0.0)
    assert np.array_equal(a.stats.trace_kept, b.stats.trace_kept)
    assert np.abs(Ta - Tb).max() <= 1e-6
    # ncclAllReduce only enqueues on the stream it is given (o3s_icp_shard_set_capturable): the second call with the same shapes
    # captures kernels AND collectives in one hipGraph, the third replays it — no collective is issued from the host any more,
    # and every bit equals the eager call's
    issued = R.o3s_rccl_collectives(comm)
    Tc = b.compute_resident(sp.T_init)
    captured = R.o3s_rccl_collectives(comm)
    Td = b.compute_resident(sp.T_init)
    assert np.array_equal(Tb, Tc) and np.array_equal(Tb, Td)
    assert np.allclose(b.stats.trace_limit, a.stats.trace_limit, rtol=2e-5, atol=0.0) and np.array_equal(b.stats.trace_kept, a.stats.trace_kept)
    assert captured > issued and R.o3s_rccl_collectives(comm) == captured, (issued, captured, R.o3s_rccl_collectives(comm))
    b.close()
    R.o3s_rccl_destroy(comm)


def test_sharded_eager_schedule_depends_on_the_input_only():
    """Every iteration of a sharded chain carries collectives the other ranks have to match, so the number of iterations a rank
    ISSUES may not depend on its handle's history (the un-sharded eager chain looks at `done` where the last call on the handle
    stopped) nor on whether its graph capture worked.  Two world-1 handles with different histories — one fresh, one that has
    just run a 15-iteration chain and a 3-iteration one — get the same new pair with a Differential checker and the capturable
    exchange: both must issue the same number of collectives, the number the chunked graph replay issues too."""
    import ctypes as C

    from open3d_slam_advanced_rss_2024_public_amd import _lib

    R = _lib.rccl_lib()
    comms = []
    for _ in range(2):
        uid = C.create_string_buffer(128)
        assert R.o3s_rccl_unique_id(uid) == 0, R.o3s_rccl_last_error()
        comm = C.c_void_p()
        assert R.o3s_rccl_create(uid, 0, 1, 0, C.byref(comm)) == 0, R.o3s_rccl_last_error()
        comms.append(comm)
    sp = syn.make_scan_pair(6000, 50000, 0.1, seed=9)
    other = syn.make_scan_pair(4000, 50000, 0.1, seed=10)
    fresh, used = ICP(IcpConfig()), ICP(IcpConfig())
    # history for `used`: a long chain (far start, never converges inside 15) and a short one, un-sharded
    assert used.init_reference(other.map_xyz, other.map_normals)
    far = other.T_init.copy()
    far[:3, 3] += 0.3
    used.compute(other.scan_xyz, other.scan_normals, far)
    used.compute(other.scan_xyz, other.scan_normals, other.T_gt)
    counts, poses, iters = [], [], []
    for h, comm in ((fresh, comms[0]), (used, comms[1])):
        assert h.init_reference(sp.map_xyz, sp.map_normals)
        h.set_reading(sp.scan_xyz, sp.scan_normals)
        h.shard_configure_rccl(sp.scan_xyz.shape[0], 0, 1, comm.value)
        per_call = []
        for _ in range(3):            # eager, capturing, replaying
            before = R.o3s_rccl_collectives(comm)
            poses.append(h.compute_resident(sp.T_init))
            per_call.append(R.o3s_rccl_collectives(comm) - before)
            iters.append(h.stats.iterations)
        counts.append(per_call)
    assert len(set(iters)) == 1 and 3 <= iters[0] < 15
    assert all(np.array_equal(poses[0], T) for T in poses[1:])
    assert counts[0][0] == counts[1][0] > 0, counts        # the eager calls issued the same collectives whatever the history
    assert counts[0][1] == counts[1][1] and counts[0][2] == counts[1][2] == 0, counts   # capture once, then replay from the graph
    for h in (fresh, used):
        h.close()
    for comm in comms:
        R.o3s_rccl_destroy(comm)
