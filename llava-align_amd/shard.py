"""Data-parallel sharding of the question list and the single result gather.

The reference shards by independent processes with contiguous ceil-chunks and no gather
(experiments/eval/MME/run_llava.py:32-40 `split_list`/`get_chunk`, --num-chunks/--chunk-idx
:261-262; each process writes its own JSONL).  Here: one process per GPU, the same contiguous
ceil-chunking — rounded to whole image groups so that the questions of one image (POPE: 6) stay
on one rank and share its ViT features and prompt-prefix KV — weights replicated, no data-path
collective, and ONE gather of the generated ids at the end of the shard (RCCL over xGMI on GPUs,
gloo in the CPU tests)."""
from __future__ import annotations

import math
from typing import List, Sequence

import torch
import torch.distributed as dist


def get_chunk(n_items: int, n_chunks: int, k: int, group: int = 1) -> range:
    """Indices of chunk k of n_chunks: contiguous, ceil-sized in units of `group` items."""
    n_groups = math.ceil(n_items / group)
    per = math.ceil(n_groups / n_chunks)
    lo, hi = min(n_groups, k * per) * group, min(n_groups, (k + 1) * per) * group
    return range(min(lo, n_items), min(hi, n_items))


def gather_tokens(local_ids: torch.Tensor, local_tokens: torch.Tensor, n_total: int, pad: int = 0) -> torch.Tensor | None:
    """local_ids [n_local] int64 question indices, local_tokens [n_local, T] int64.  Returns on every rank the
    [n_total, T] matrix of generated ids (one all_gather of a few hundred bytes per question)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        out = torch.full((n_total, local_tokens.shape[1]), pad, dtype=torch.long, device=local_tokens.device)
        out[local_ids] = local_tokens
        return out
    world = dist.get_world_size()
    dev = local_tokens.device
    T = local_tokens.shape[1]
    n_local = torch.tensor([local_ids.numel()], dtype=torch.long, device=dev)
    counts = [torch.zeros(1, dtype=torch.long, device=dev) for _ in range(world)]
    dist.all_gather(counts, n_local)
    cap = int(max(c.item() for c in counts))
    buf = torch.full((cap, T + 1), -1, dtype=torch.long, device=dev)          # column 0: question index, -1 = padding row
    buf[: local_ids.numel(), 0] = local_ids
    buf[: local_ids.numel(), 1:] = local_tokens
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    out = torch.full((n_total, T), pad, dtype=torch.long, device=dev)
    for b in bufs:
        ok = b[:, 0] >= 0
        out[b[ok, 0]] = b[ok, 1:]
    return out
