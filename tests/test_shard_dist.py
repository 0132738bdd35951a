"""Sharding + the single result gather, world_size 2 over gloo on CPU (the N>1 path of bench.py / the eval
drivers uses the same functions over RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from llava_align_amd.shard import gather_tokens, get_chunk


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
    ret[rank] = bool(torch.equal(out, want))
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
