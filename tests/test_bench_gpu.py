"""bench.py end to end on the GPU box: the N=1 contract line (roofline + cpu_baseline objects, whole-step figure), and the
self-spawned N=2 launch (`python bench.py --gpus 2`, WORLD_SIZE unset) rehearsed on the one GPU with MSC_BENCH_BACKEND=gloo —
both ranks share cuda:0 and exchange gradients through host memory, the same Trainer code path the driver's RCCL runs take."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_self_spawned_ranks_on_one_gpu(cuda):
    out = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--no-cpu-baseline"], {"MSC_BENCH_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 8
    assert out["dp_params_in_sync"] is True and out["rccl_ranks"] == 2 and out["scaling"] == "weak"
    assert [p[0] for p in out["dp_schedule"]] == ["upper", "mid", "rest"]
    assert out["value"] > 0 and out["roofline"]["frac"] > 0


def test_bench_contract_line_single_gpu(cuda):
    out = _bench(["--steps", "3", "--warmup", "2", "--batch", "8", "--no-cpu-baseline"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "build_id"):
        assert k in out, k
    rf = out["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    # the whole-step figure uses the driver-checkable clock: FLOPs of the contraction kernel / ms_per_step
    assert abs(rf["whole_step_tflops"] - rf["algorithmic_gflop_per_step"] / out["ms_per_step"]) < 1e-6 * rf["whole_step_tflops"] + 1e-9
    assert rf["whole_step_frac"] <= rf["frac"] * 1.05
    assert out["vs_baseline"] is None and out["dtype"] == "bf16" and "workload" in out["config"]
