"""bench.py as the driver runs it: one JSON line on stdout that fits the driver's capture, and `--gpus N` starting its own
ranks when no launcher is around it (round-4 review items 1 and 3)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")
SHORT = ["--steps", "3", "--warmup", "2", "--no-presets", "--no-epoch", "--no-cpu-baseline"]


def _env(**kw):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **kw)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "RR_F16X2"):
        env.pop(k, None)
    return env


def _run(args, env):
    p = subprocess.run([sys.executable, BENCH] + args, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                            # ONE line on stdout, nothing else
    assert len(lines[0].encode()) < 6144, len(lines[0])
    return json.loads(lines[0]), p.stderr


def test_one_gpu_line_is_short_complete_and_on_the_exact_arithmetic():
    line, err = _run(["--gpus", "1"] + SHORT, _env())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype",
              "data", "config", "roofline", "roofline_gather"):
        assert line.get(k) is not None, k
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["dtype"] == "f32" and "three EXACT bf16" in line["gemm_arithmetic"]
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["products_per_multiply"] == 6 and "linear_split_kernel" in r["kernel"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["avg_launch_us"] > 0
    assert line["f16x2_path"]["dtype"] != "f32" and line["f16x2_path"]["queries_per_s"] > 0     # the narrower form is labelled as such
    assert line["f32_mfma_path"]["queries_per_s"] > 0
    assert abs(line["value"] - 64 * 1e3 / line["ms_per_step"]) < 0.01 * line["value"]
    with open(os.path.join(REPO, line["detail"])) as f:
        detail = json.load(f)
    assert "kernels" in detail and "kernels_isolated" in detail and "detail kernels" in err


def test_gpus_2_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` as the driver may run it: the parent starts torch.distributed.run as a child before touching
    the GPU and relays rank 0's line.  On a one-GPU box the two ranks share GPU 0 and talk over gloo."""
    import torch
    extra = {} if torch.cuda.device_count() >= 2 else dict(RR_SINGLE_DEVICE="1", RR_DIST_BACKEND="gloo")
    line, _ = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-presets", "--no-epoch", "--no-cpu-baseline", "--no-f32-path",
                    "--no-fwd-only"], _env(**extra))
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["scaling"] == "weak"
    assert len(line["dp"]["rccl_ranks"]) == 2 and {r["rank"] for r in line["dp"]["rccl_ranks"]} == {0, 1}
    assert line["dp"]["allreduce_us"]["calls"] >= 2 and line["value"] > 0
