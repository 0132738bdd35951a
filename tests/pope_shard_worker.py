"""Worker of tests/test_drivers_gpu.py::test_run_pope_sharded_over_two_ranks_equals_one_rank: started by torch.distributed.run (one
process per rank; on the single-GPU test box both ranks use device VDD_FORCE_DEVICE over gloo, on a node one GPU each over RCCL).
Runs pope_driver.run_pope on a seeded synthetic POPE list with the tiny engine and writes what THIS rank returned to
<out>.rank<k>.json; rank 0 also writes the answers JSONL (the driver's own rank-0 write)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402


def questions(n_img=7, per_img=3):
    out = []
    for i in range(n_img * per_img):
        im = (i * 5) % n_img                                   # images interleaved in file order: the driver's image sort matters
        out.append({"question_id": 1000 + i, "image": f"im{im}.jpg", "text": f"Is there a thing number {i} in the image?",
                    "label": "yes" if i % 3 else "no"})
    return out


def encode(text, with_image):
    ids = [1, 11, 12, 13] + ([-200] if with_image else []) + [3 + (sum(map(ord, w)) % 900) for w in text.split()]
    return ids


def decode(ids):
    return " ".join(("yes" if t % 2 else "no") for t in ids)


def load_image(name):
    return torch.randn(3, 56, 56, generator=torch.Generator().manual_seed(int(name[2:-4])))


def main(out_path):
    from llava_align_amd.engine import LlavaWeights, VddLlavaEngine, preset
    from llava_align_amd.pope_driver import run_pope
    from llava_align_amd.shard import init_from_env
    rank, world, dev = init_from_env()
    # nothing set by hand: cd_greedy makes run_pope decode in batch-invariant mode (shard.resolve_batch_invariant), in which a row's logits
    # do not depend on which other questions share its batch (the chunk of a rank is a different batch than the whole list) nor on a
    # timing-based tuner pick - so 2 ranks and 1 rank must agree token for token
    cfg = preset("tiny")
    eng = VddLlavaEngine(cfg, weights=LlavaWeights.random(cfg, dev, seed=3, std=0.06, dtype=torch.float16), device=dev, use_graph=True)
    res = run_pope(eng, questions(), encode, decode, load_image, answers_path=out_path + ".jsonl", batch_questions=8, max_new_tokens=6,
                   use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, cd_greedy=True, eos_token_id=None)
    with open(f"{out_path}.rank{rank}.json", "w") as f:
        json.dump({"rank": res["rank"], "world": res["world"], "answers": res["answers"], "scores": res["scores"]}, f)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
