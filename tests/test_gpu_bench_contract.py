"""bench.py's output contract on the GPU box: one JSON line with the fields the driver reads (metric / value / unit /
n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload), the
`roofline` object of the dominant kernel measured live with HIP events, and -- at N=1 -- the `cpu_baseline` object with the
parity figures of the metric's second half; the training workload's line; and the N=2 launch path (ranks started by
bench.py itself, both on the one test GPU over gloo: a rehearsal of the plumbing, not a scaling number)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=e, capture_output=True,
                       text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout: %r" % p.stdout[-500:]
    return json.loads(lines[0])


def test_default_workload_line():
    d = _run(["--steps", "4", "--warmup", "2", "--no-alt"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "configs[1]" in d["config"]["workload"] and "model" not in d["config"]
    assert d["value"] > 50 and abs(d["value"] * d["ms_per_step"] - 1000.0) < 1.0
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.3 < r["frac"] < 1.0
    assert r["launches_timed"] == 4 and r["traffic"] > 8e7 and "profiles/" in r["traffic_source"]
    # the loop-invariant work is hoisted by default: the line says so and prices the executed flops beside the algorithmic
    assert "cond_projection" in d["config"] and 0.3 < r["executed"]["frac"] < r["frac"]
    assert r["executed"]["flop_per_frame"] == 23805952 - 0.75 * 2621440
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    par = c["parity"]
    assert par["x0_pred"]["max_abs_over_max_ref"] < par["tolerance"] == 1e-3
    assert par["final_mel"]["max_abs_over_max_ref"] < 1e-3


def test_both_legs_and_the_per_step_headline():
    d = _run(["--steps", "8", "--warmup", "4", "--no-cpu-baseline"])          # with the alt legs
    p = d["projection_in_every_step"]
    assert p["unit"] == "steps/s" and 0.5 * d["value"] < p["value"] < 1.02 * d["value"]
    assert "executed" not in p["roofline"] and p["roofline"]["launches_timed"] == 8
    assert d["alt"]["dtype"].startswith("bf16x3")
    q = _run(["--steps", "8", "--warmup", "4", "--no-cpu-baseline", "--no-alt", "--project-per-step"])
    assert "cond_projection" not in q["config"] and "executed" not in q["roofline"]
    assert "projection_in_every_step" not in q


def test_train_workload_line():
    d = _run(["--workload", "train", "--steps", "2", "--warmup", "1"])
    assert "configs[3]" in d["config"]["workload"] and d["n_gpus"] == 1 and d["steps"] == 2
    assert d["unit"] == "steps/s" and d["value"] > 10 and d["dtype"] == "f32"
    assert "comm_exposed_ms" not in d          # no process group at N=1 unless MG_BENCH_EXCHANGE=1


def test_two_ranks_started_by_bench_itself():
    d = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-alt", "--no-cpu-baseline"],
             env={"MG_BENCH_SHARE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert "cpu_baseline" not in d             # rank 0 at N=1 only
    t = _run(["--gpus", "2", "--workload", "train", "--steps", "2", "--warmup", "1"], env={"MG_BENCH_SHARE_GPU": "1"})
    assert t["n_gpus"] == 2 and t["backend"] == "gloo" and "comm_exposed_ms" in t and t["grad_bytes"]["G_early_chunk"] > 0
