"""Sharding + the single result gather, world_size 2 over gloo on CPU (the N>1 path of bench.py / the eval
drivers uses the same functions over RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from llava_align_amd.shard import gather_results, gather_tokens, get_chunk


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


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_total, T = 11, 5
    mine = torch.tensor(list(get_chunk(n_total, world, rank, group=3)))
    toks = (mine[:, None] * 100 + torch.arange(T)[None]).long()
    out = gather_tokens(mine, toks, n_total)
    want = (torch.arange(n_total)[:, None] * 100 + torch.arange(T)[None]).long()
    ok = bool(torch.equal(out, want))
    # ranks whose batches stopped at EOS after different numbers of steps hold different T: padded to the longest
    t_r = T - rank
    out2 = gather_tokens(mine, toks[:, :t_r], n_total, pad=-7)
    want2 = want.clone()
    other = torch.tensor(list(get_chunk(n_total, world, 1, group=3)))
    want2[other, T - 1:] = -7
    # the full section-8(e) payload {qid, n_tokens, tokens, top10_tok, top10_prob} in one all_gather
    tt = (mine[:, None] * 7 + torch.arange(10)[None]).long()
    tp = (mine[:, None].float() * 0.01 + torch.arange(10)[None].float() * 1e-3)
    res = gather_results(mine, toks[:, :t_r], torch.full((mine.numel(),), t_r), tt, tp, n_total, pad=-7)
    allq = torch.arange(n_total)
    ok3 = (torch.equal(res["tokens"], want2) and torch.equal(res["top_tok"], (allq[:, None] * 7 + torch.arange(10)[None]).long())
           and torch.equal(res["top_prob"], allq[:, None].float() * 0.01 + torch.arange(10)[None].float() * 1e-3)
           and res["n_tokens"][other].tolist() == [T - 1] * other.numel())
    ret[rank] = ok and bool(torch.equal(out2, want2)) and ok3
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
    out = gather_tokens(ids, toks, 3, pad=9)
    assert out.tolist() == [[7, 8], [9, 9], [5, 6]]
