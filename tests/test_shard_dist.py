"""Sharding + the single result gather, world_size 2 over gloo on CPU (the N>1 path of bench.py / the eval
drivers uses the same functions over RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from llava_align_amd.shard import ShardPlan, gather_results, get_chunk


def test_chunks_partition_and_keep_image_groups_together():
    for n, w, g in ((3000, 8, 6), (90, 8, 1), (500, 3, 6), (7, 4, 6), (0, 2, 6)):
        seen = []
        for k in range(w):
            r = get_chunk(n, w, k, group=g)
            seen += list(r)
            if len(r) and g > 1:
                assert r.start % g == 0                         # a chunk starts on an image-group boundary
        assert seen == list(range(n))                           # contiguous, disjoint, complete
    assert list(get_chunk(10, 3, 0)) == [0, 1, 2, 3] and list(get_chunk(10, 3, 2)) == [8, 9]     # reference ceil-chunking


class _StubEngine:
    """generate() of the native engine, replaced by a pure function of the prompt ids: what a driver's sharding must not change."""
    device = torch.device("cpu")

    def clear_image_cache(self):
        pass

    def generate(self, ids, images=None, max_new_tokens=4, n_top=0, eos_token_id=None, **kw):
        import types
        Q = len(ids)
        base = torch.tensor([int(r.sum()) % 89 for r in ids])
        T = max_new_tokens if eos_token_id is None else min(max_new_tokens, 2)          # an "EOS" run stops early: fewer columns
        toks = (base[:, None] + torch.arange(T)[None]) % 97 + 3
        tt = (base[:, None] * 3 + torch.arange(10)[None]) % 97 + 3
        tp = torch.softmax(-torch.arange(10.0)[None] * (1 + base[:, None].float() / 89), -1)
        return types.SimpleNamespace(tokens=toks, top_tok=tt, top_prob=tp)


def _pope_inputs():
    qs = [{"question_id": 100 + i, "image": f"img{(i * 7) % 5}.jpg", "text": f"Is there a thing {i}?", "label": "yes" if i % 2 else "no"}
          for i in range(23)]
    encode = lambda text, with_image: [1, 5] + ([-200] if with_image else []) + [3 + (ord(c) % 50) for c in text[-6:]]
    decode = lambda ids: " ".join(("yes" if t % 2 else "no") for t in ids)
    load_image = lambda name: torch.full((3, 2, 2), float(int(name[3])))
    return qs, encode, decode, load_image


def _driver_check(rank, world):
    """run_pope over 2 ranks == run_pope on one: same answers on EVERY rank, the file written once (rank 0)."""
    import json
    import tempfile
    from llava_align_amd.pope_driver import run_pope
    qs, encode, decode, load_image = _pope_inputs()
    path = os.path.join(tempfile.gettempdir(), f"vdd_pope_shard_{os.environ['MASTER_PORT']}.jsonl")
    if rank == 0 and os.path.exists(path):
        os.remove(path)
    dist.barrier()
    kw = dict(batch_questions=4, max_new_tokens=5, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, eos_token_id=2, pad_token_id=0)
    res = run_pope(_StubEngine(), qs, encode, decode, load_image, answers_path=path, **kw)                      # rank / world from the group
    one = run_pope(_StubEngine(), qs, encode, decode, load_image, rank=0, world=1, **kw)
    dist.barrier()
    ok = res["world"] == world and res["answers"] == one["answers"] and res["scores"] == one["scores"]
    lines = [json.loads(l) for l in open(path)]
    ok = ok and [l["question_id"] for l in lines] == [q["question_id"] for q in qs] and lines[3]["text"] == one["answers"][3]["text"]
    plan = ShardPlan(sorted(q["image"] for q in qs), rank, world)
    return ok and 0 < len(plan.mine) < len(qs)


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_total, T = 11, 5
    mine = torch.tensor(list(get_chunk(n_total, world, rank, group=3)))
    toks = (mine[:, None] * 100 + torch.arange(T)[None]).long()
    want = (torch.arange(n_total)[:, None] * 100 + torch.arange(T)[None]).long()
    ok = True
    # ranks whose batches stopped at EOS after different numbers of steps hold different T: padded to the longest
    t_r = T - rank
    want2 = want.clone()
    other = torch.tensor(list(get_chunk(n_total, world, 1, group=3)))
    want2[other, T - 1:] = -7
    # the full section-8(e) payload {qid, n_tokens, tokens, top10_tok, top10_prob} in one all_gather
    tt = (mine[:, None] * 7 + torch.arange(10)[None]).long()
    tp = (mine[:, None].float() * 0.01 + torch.arange(10)[None].float() * 1e-3)
    allq = torch.arange(n_total)
    cap = max(len(get_chunk(n_total, world, k, group=3)) for k in range(world))
    ok3 = True
    for kw in (dict(capacity=cap, width=T), dict()):        # the shard plan known up front: ONE collective; unknown: + one all_reduce(MAX)
        res = gather_results(mine, toks[:, :t_r], torch.full((mine.numel(),), t_r), tt, tp, n_total, pad=-7, **kw)
        ok3 = ok3 and (torch.equal(res["tokens"], want2) and torch.equal(res["top_tok"], (allq[:, None] * 7 + torch.arange(10)[None]).long())
                       and torch.equal(res["top_prob"], allq[:, None].float() * 0.01 + torch.arange(10)[None].float() * 1e-3)
                       and res["n_tokens"][other].tolist() == [T - 1] * other.numel() and bool((res["count"] == 1).all()))
    # one collective per gather when the plan is known: count the calls that reach the backend
    calls = []
    real = (dist.all_gather_into_tensor, dist.all_gather, dist.all_reduce)
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append("agt"), real[0](*a, **k))[1]
    dist.all_gather = lambda *a, **k: (calls.append("ag"), real[1](*a, **k))[1]
    dist.all_reduce = lambda *a, **k: (calls.append("ar"), real[2](*a, **k))[1]
    gather_results(mine, toks, torch.full((mine.numel(),), T), tt, tp, n_total, capacity=cap, width=T)
    dist.all_gather_into_tensor, dist.all_gather, dist.all_reduce = real
    ok4 = calls in (["agt"], ["agt", "ag"])                   # (the list form only where the backend lacks the flat one)
    ok5 = _driver_check(rank, world)
    ret[rank] = ok and ok3 and ok4 and ok5
    dist.barrier()
    dist.destroy_process_group()


def test_gather_world_size_2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_gather_single_process():
    ids = torch.tensor([2, 0])
    toks = torch.tensor([[5, 6], [7, 8]])
    out = gather_results(ids, toks, torch.tensor([2, 2]), torch.zeros(2, 10, dtype=torch.long), torch.zeros(2, 10), 3, pad=9)["tokens"]
    assert out.tolist() == [[7, 8], [9, 9], [5, 6]]
