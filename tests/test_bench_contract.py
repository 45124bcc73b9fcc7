"""bench.py prints ONE JSON line with the fields the driver reads (small geometry, on the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--grid", "96", "--nrhs", "128", "--steps", "2", "--warmup", "1",
                          "--cpu-rhs", "4"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f64" and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["value"] > 0 and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    assert d["check"]["LLt_residual_col0"] <= 1e-10
