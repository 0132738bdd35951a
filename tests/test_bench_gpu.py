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


def test_bench_strong_scaling_mode_splits_one_list_over_two_ranks():
    """`--strong N`: one seeded question list, rank k takes shard.get_chunk(N, world, k, group=6) - the code path the eval drivers use on
    a node - EOS on, ragged results gathered once; rank 0 asserts that the gathered question ids cover 0..N-1 exactly once."""
    line = _run(["--gpus", "2", "--model", "tiny", "--strong", "42", "--questions", "12", "--steps", "1", "--warmup", "1", "--no-baselines"],
                {"VDD_FORCE_DEVICE": "0", "VDD_DIST_BACKEND": "gloo"})
    assert line["scaling"] == "strong" and line["n_gpus"] == 2 and line["config"]["questions_total"] == 42
    assert line["config"]["questions_rank0"] == 24                       # ceil(7 image groups / 2) = 4 groups of 6
    assert line["gathered_question_ids_cover_the_list_once"] and 1.0 <= line["mean_answer_tokens"] <= 2.0
    one = _run(["--model", "tiny", "--strong", "42", "--questions", "12", "--steps", "1", "--warmup", "0", "--no-baselines"], {})
    assert one["n_gpus"] == 1 and one["config"]["questions_rank0"] == 42


def test_bench_result_gather_through_rccl_at_world_size_one():
    """The one thing a single-GPU box can do for the N > 1 path: run the SAME process-group code through RCCL (backend "nccl" with
    device_id, barrier, all_reduce of the timing, the all_gather of the section-8(e) payload on device tensors) at world size 1."""
    line = _run(["--model", "tiny", "--questions", "12", "--steps", "1", "--warmup", "1", "--no-baselines"],
                {"VDD_FORCE_DIST": "1", "VDD_DIST_BACKEND": "nccl", "MASTER_PORT": "29541", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert line["collective_backend"] == "nccl" and line["n_gpus"] == 1 and line["value"] > 0
