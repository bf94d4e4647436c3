"""bench.py is the measurement contract (one JSON line per run): run it for real, small, on the GPU box -- every
configuration, the dropout row, the fp8 mode and the CPU-baseline leg -- and check the line's shape."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(*args, timeout=600):
    env = dict(os.environ)
    env.pop("FAVIT_DP_FORCE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _check(j, steps, warmup, dtype="bf16"):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["unit"] == "images/sec" and j["n_gpus"] == 1 and j["steps"] == steps and j["warmup"] == warmup
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["dtype"] == dtype and j["data"] == "synthetic" and "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 0 and j["ms_per_step"] > 0
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("mfma", "hbm") and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    import math
    assert math.isfinite(j["loss"])
    assert j["knobs"] == {k: v for k, v in os.environ.items() if k.startswith("FAVIT_") and k != "FAVIT_DP_FORCE"}


@pytest.mark.parametrize("cfg,extra", [("cfg2", ["--batch", "16"]), ("cfg2", ["--batch", "16", "--dropout", "0.1"]),
                                       ("cfg2", ["--batch", "16", "--dropout", "0.1", "--attn-dropout", "0.1", "--embed-dropout", "0.1"]),
                                       ("cfg1", []), ("cfg3", ["--batch", "16"]), ("cfg3", ["--batch", "16", "--slic"]),
                                       ("cfg5", ["--batch", "16"]), ("cfg2", ["--batch", "16", "--dtype", "fp32"]),
                                       ("cfg4", ["--batch", "2", "--dtype", "fp8"])])
def test_bench_line(cfg, extra):
    j = _run("--config", cfg, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", *extra)
    _check(j, 3, 1, dtype="fp8" if "fp8" in extra else "fp32" if "fp32" in extra else "bf16")
    assert j["config"]["baseline_config"] == cfg
    if "--slic" in extra:
        assert j["slic_inclusive"]["ms_per_step"] > j["ms_per_step"] and j["slic_inclusive"]["images_per_sec"] > 0
    if cfg == "cfg5":
        assert j["config"]["hip_graph"] is True and "R = 16 / R = 15" in j["config"]["workload"]
    if "--dropout" in extra:
        # the reference's setting is dropout 0.1 with attn / embed dropout 0.0 (main.py:106-111): --dropout sets only the first
        assert j["config"]["dropout"] == 0.1
        want = 0.1 if "--attn-dropout" in extra else 0.0
        assert j["config"]["attn_dropout"] == want and j["config"]["embed_dropout"] == want


def test_bench_refuses_to_report_a_non_finite_run(tmp_path):
    """A poisoned run is not a measurement: bench.py exits non-zero WITHOUT a metric line, names the first non-finite
    tensor and the AdamW launch it appeared at (the library's health word), and leaves the record under gpurun_out/."""
    env = dict(os.environ, FAVIT_BENCH_TEST_POISON_STEP="2")
    env.pop("FAVIT_POISON", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "cfg1", "--steps", "4", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 3, (out.returncode, out.stderr[-1500:])
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "NON-FINITE" in out.stderr
    line = [l for l in out.stderr.splitlines() if "NON-FINITE" in l][0]
    rep = json.loads(line.split("NON-FINITE values in the step: ", 1)[1].rsplit("  (written to", 1)[0])
    # 1 warm-up step + timed step 0, 1 are clean; the poison goes in before timed step 2 = the 4th AdamW launch round
    per = rep["adamw_launches_per_step"]
    assert rep["adamw_launch_of_first_bad_gradient_or_parameter"] in range(3 * per + 1, 4 * per + 2)
    assert any(t.get("buffer") == "flat_p" for t in rep["tensors"])
    assert rep["knobs"] == {"FAVIT_BENCH_TEST_POISON_STEP": "2"}
    path = line.rsplit("(written to ", 1)[1].rstrip(")")
    assert os.path.exists(path)
    os.remove(path)


def test_bench_cpu_baseline_leg():
    """The oracle-timed CPU baseline (kind "port") on the small configuration."""
    j = _run("--config", "cfg1", "--steps", "2", "--warmup", "1")
    _check(j, 2, 1)
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1


def test_bench_two_ranks_gloo_on_one_gpu():
    """The multi-rank flow of bench.py (rendezvous on 127.0.0.1, barriers, max-over-ranks timing, the collective
    GEMM-trace steps, rank-0 JSON) with two ranks sharing the box's one GPU; gloo stands in for RCCL, which needs one
    GPU per rank."""
    env = dict(os.environ)
    env.pop("FAVIT_DP_FORCE", None)
    port = 36000 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
           "--batch", "16", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 32 and j["config"]["parallelism"] == "dp2"
    assert j["scaling"] == "weak" and j["value"] > 0 and "cpu_baseline" not in j
