"""N > 1 path on CPU: world_size-2 gloo run of the pair-sharding logic (parallel.run_pairs_sharded).  The ICP runner in
this CPU test is the oracle (test infrastructure); on the GPU box the same code path uses parallel.gpu_runner."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from open3d_slam_advanced_rss_2024_public_amd import parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_indices_cover_every_unit_once():
    for n in (0, 1, 7, 64):
        for world in (1, 2, 3, 8):
            seen = sorted(u for r in range(world) for u in parallel.shard_indices(n, world, r))
            assert seen == list(range(n))
            sizes = [len(parallel.shard_indices(n, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, os.environ["REPO_ROOT"])
    from open3d_slam_advanced_rss_2024_public_amd import parallel, synthetic as syn
    from oracle import oracle as orc

    def oracle_runner(pairs):
        out = []
        for p in pairs:
            o = orc.OracleIcp(orc.OracleConfig(), threads=1)
            o.init_reference(p["map_xyz"], p["map_normals"])
            T, code = o.compute(p["scan_xyz"], p["scan_normals"], p["T_init"], raise_on_error=False)
            out.append((T, code, o.stats.iterations))
        return out

    dist.init_process_group(backend="gloo")
    pairs = []
    for k in range(5):
        sp = syn.make_scan_pair(1500, 12000, 0.1, seed=10 + k)
        pairs.append(dict(map_xyz=sp.map_xyz, map_normals=sp.map_normals, scan_xyz=sp.scan_xyz, scan_normals=sp.scan_normals,
                          T_init=sp.T_init))
    pairs[3]["scan_xyz"] = pairs[3]["scan_xyz"] + 500.0   # this pair must fail with NO_MATCHES on whichever rank owns it
    res = parallel.run_pairs_sharded(pairs, oracle_runner, dist=dist)
    ref = oracle_runner(pairs) if dist.get_rank() == 0 else None
    if dist.get_rank() == 0:
        ok = all((a[0] is None) == (b[0] is None) and a[1] == b[1] and a[2] == b[2] and
                 (a[0] is None or np.allclose(a[0], b[0], atol=0)) for a, b in zip(res, ref))
        print(json.dumps({"ok": bool(ok), "statuses": [r[1] for r in res], "n": len(res)}))
    dist.barrier()
    dist.destroy_process_group()
""")


def test_world2_gloo_sharded_pairs(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, REPO_ROOT=ROOT, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    import json

    r = json.loads(line)
    assert r["ok"] and r["n"] == 5
    assert r["statuses"][3] == 5 and all(s == 0 for k, s in enumerate(r["statuses"]) if k != 3)
