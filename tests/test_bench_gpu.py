"""bench.py as the driver launches it: `--gpus N` must become N ranks by itself (here 2 ranks sharing the one GPU of the test
box over gloo: VDD_FORCE_DEVICE / VDD_DIST_BACKEND), report the real world size and gather the full per-question payload."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra):
    env = dict(os.environ, **env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus_2_launches_two_ranks_itself():
    line = _run(["--gpus", "2", "--model", "tiny", "--questions", "12", "--steps", "1", "--warmup", "1", "--no-baselines"],
                {"VDD_FORCE_DEVICE": "0", "VDD_DIST_BACKEND": "gloo"})
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["scaling"] == "weak"
    assert line["value"] > 0 and abs(line["value"] - 2 * line["tokens_per_s_per_gpu"]) <= 0.02 * line["value"]
    assert line["pope_eos"]["mean_answer_tokens"] <= 2.0 and line["pope_eos"]["decode_steps_run"] <= 4


def test_bench_single_rank_line_has_the_contract_fields():
    line = _run(["--model", "tiny", "--questions", "12", "--steps", "1", "--warmup", "1", "--no-baselines"], {})
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in line
    assert line["n_gpus"] == 1 and "workload" in line["config"]
