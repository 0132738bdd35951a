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
    meta = torch.tensor([local_ids.numel(), local_tokens.shape[1]], dtype=torch.long, device=dev)
    metas = [torch.zeros(2, dtype=torch.long, device=dev) for _ in range(world)]
    dist.all_gather(metas, meta)
    cap = int(max(c[0].item() for c in metas))
    T = int(max(c[1].item() for c in metas))               # a rank whose batch hit EOS early returns fewer columns: pad to the longest
    buf = torch.full((cap, T + 1), -1, dtype=torch.long, device=dev)          # column 0: question index, -1 = padding row
    buf[:, 1:] = pad
    buf[: local_ids.numel(), 0] = local_ids
    buf[: local_ids.numel(), 1: 1 + local_tokens.shape[1]] = local_tokens
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    out = torch.full((n_total, T), pad, dtype=torch.long, device=dev)
    for b in bufs:
        ok = b[:, 0] >= 0
        out[b[ok, 0]] = b[ok, 1:]
    return out


def gather_results(local_ids: torch.Tensor, tokens: torch.Tensor, n_tokens: torch.Tensor, top_tok: torch.Tensor | None,
                   top_prob: torch.Tensor | None, n_total: int, pad: int = 0) -> dict:
    """The per-question payload of SURVEY.md section 8(e) in ONE all_gather: {qid, n_tokens, tokens[T], top10_tok, top10_prob}.
    Everything rides in one int64 matrix (the fp32 probabilities bit-cast into it), a few hundred bytes per question.
    Returns, on every rank, tensors ordered by question index."""
    dev = tokens.device
    n_local, T = tokens.shape
    k = 0 if top_tok is None else int(top_tok.shape[1])
    cols = [local_ids.view(-1, 1).long(), n_tokens.view(-1, 1).long(), tokens.long()]
    if k:
        cols += [top_tok.long(), top_prob.float().contiguous().view(torch.int32).long()]
    packed = torch.cat(cols, dim=1)
    import os
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and os.environ.get("VDD_FORCE_DIST") != "1"):
        blocks = [packed]
    else:
        world = dist.get_world_size()
        meta = torch.tensor([n_local, T, k], dtype=torch.long, device=dev)
        metas = [torch.zeros(3, dtype=torch.long, device=dev) for _ in range(world)]
        dist.all_gather(metas, meta)
        cap, Tm = int(max(m[0].item() for m in metas)), int(max(m[1].item() for m in metas))
        if any(int(m[2].item()) != k for m in metas):
            raise ValueError("ranks disagree on the number of top-n entries")
        buf = torch.full((cap, 2 + Tm + 2 * k), -1, dtype=torch.long, device=dev)       # qid -1 = padding row
        buf[:, 2: 2 + Tm] = pad
        buf[:n_local, :2] = packed[:, :2]
        buf[:n_local, 2: 2 + T] = packed[:, 2: 2 + T]
        if k:
            buf[:n_local, 2 + Tm:] = packed[:, 2 + T:]
        bufs = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(bufs, buf)
        blocks, T = bufs, Tm
    out = {"tokens": torch.full((n_total, T), pad, dtype=torch.long, device=dev), "n_tokens": torch.zeros(n_total, dtype=torch.long, device=dev)}
    if k:
        out["top_tok"] = torch.full((n_total, k), -1, dtype=torch.long, device=dev)
        out["top_prob"] = torch.zeros(n_total, k, dtype=torch.float32, device=dev)
    out["count"] = torch.zeros(n_total, dtype=torch.long, device=dev)       # how many ranks delivered each question (a partition: all 1)
    for b in blocks:
        ok = b[:, 0] >= 0
        q = b[ok, 0]
        out["count"].index_add_(0, q, torch.ones_like(q))
        out["n_tokens"][q] = b[ok, 1]
        out["tokens"][q] = b[ok, 2: 2 + T]
        if k:
            out["top_tok"][q] = b[ok, 2 + T: 2 + T + k]
            out["top_prob"][q] = b[ok, 2 + T + k:].to(torch.int32).view(torch.float32)
    return out
