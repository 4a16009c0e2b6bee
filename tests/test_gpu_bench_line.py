"""The one JSON line `bench.py` prints: the contract's fields, and that what `config` says is what `value` was measured on."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_bench_line_describes_the_run_that_was_timed():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu"], capture_output=True,
                         text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout            # exactly one line on stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["scaling"] == "weak"
    c = d["config"]
    assert c["workload"].startswith("C2") and c["scan_points"] == 100_000 and c["map_points"] == 2_000_000 and c["iterations_per_step"] == 50
    # ONE pair per GPU was timed: the "8 pairs in flight" figure is an `extra`, it must not leak into the description of `value`
    assert c["pairs_per_gpu"] == 1 and c["parallelism"].startswith("1 independent scan/map pair, 1 per GPU")
    assert abs(d["value"] - 50 * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3)) <= 1e-3 * d["value"]
    assert d["extra"]["batched_on_one_gpu"]["pairs_in_flight"] == 8 and d["extra"]["batched_on_one_gpu"]["value"] > d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["kernel"] == "k_match2" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert {k_["kernel"] for k_ in d["roofline_kernels"]} == {"k_match2", "k_classify", "k_sel_ne", "k_solve"}
