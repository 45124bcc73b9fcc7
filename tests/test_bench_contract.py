"""bench.py prints ONE JSON line with the fields the driver reads (small geometry, on the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env=None, timeout=900):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + argv, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=e)


def _one_line(out):
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """--gpus must agree with the launcher's WORLD_SIZE: no line for the wrong size (exits before anything touches a GPU)."""
    out = _run(["--gpus", "4"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode == 2 and not out.stdout.strip() and "WORLD_SIZE" in out.stderr


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    d = _one_line(_run(["--grid", "96", "--nrhs", "128", "--steps", "2", "--warmup", "1", "--cpu-rhs", "4"]))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "roofline_chol", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f64" and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["value"] > 0 and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] is None and len(r["kernels_sha"]) == 16  # no committed counter pass speaks for this small geometry
    rc = d["roofline_chol"]
    assert rc["bound"] in ("hbm", "mfma") and rc["achieved"] > 0 and abs(rc["frac"] - rc["achieved"] / rc["peak"]) < 1e-12
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    assert c["schol_s"] > 0 and c["chol_s"] > 0 and c["nproc"] >= 1
    k = d["check"]
    assert k["columns_checked"] == 128 and k["residual_max_all_columns"] <= 1e-10 and len(k["residual_spread_columns"]) == 8
    assert k["LLt_residual_col0"] <= 1e-10


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal_self_spawned():
    """`bench.py --gpus 2` without a launcher starts its two ranks itself; on a one-GPU box both sit on cuda:0 over gloo
    (RSP_BENCH_REHEARSAL): broadcast of the factor, barriers, max over ranks, one line from rank 0 with the CPU baseline."""
    d = _one_line(_run(["--gpus", "2", "--grid", "64", "--nrhs", "64", "--steps", "2", "--warmup", "1", "--cpu-rhs", "2", "--skip-extras"],
                       env={"RSP_BENCH_REHEARSAL": "1"}))
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and len(d["ms_per_step_per_rank"]) == 2 and "cpu_baseline" in d
    assert d["check"]["residual_max_all_columns"] <= 1e-10


@pytest.mark.gpu
def test_bench_config_modes_c3_strong_and_c5():
    d = _one_line(_run(["--config", "c3", "--scaling", "strong", "--grid", "16", "--nrhs", "128", "--steps", "2", "--warmup", "1",
                        "--cpu-rhs", "0", "--skip-extras"]))
    assert d["scaling"] == "strong" and d["config"]["rhs_total"] == 128 and d["config"]["columns_per_rank"] == [128]
    assert d["check"]["residual_max_all_columns"] <= 1e-10
    d = _one_line(_run(["--gpus", "2", "--config", "c3", "--scaling", "strong", "--grid", "16", "--nrhs", "128", "--steps", "2",
                        "--warmup", "1", "--cpu-rhs", "0", "--skip-extras"], env={"RSP_BENCH_REHEARSAL": "1"}))
    assert d["n_gpus"] == 2 and d["config"]["columns_per_rank"] == [64, 64]
    d = _one_line(_run(["--config", "c5", "--systems", "64", "--steps", "2", "--warmup", "1"]))
    assert d["unit"] == "systems/s" and d["value"] > 0 and d["check"]["residual_max_sampled"] <= 1e-10
    d = _one_line(_run(["--gpus", "2", "--config", "c5", "--systems", "64", "--steps", "2", "--warmup", "1"], env={"RSP_BENCH_REHEARSAL": "1"}))
    assert d["n_gpus"] == 2 and d["config"]["systems_rank0"] == 32
