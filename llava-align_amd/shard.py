"""Data-parallel sharding of the question list and the single result gather.

The reference shards by independent processes with contiguous ceil-chunks and no gather
(experiments/eval/MME/run_llava.py:32-40 `split_list`/`get_chunk`, --num-chunks/--chunk-idx
:261-262; each process writes its own JSONL).  Here: one process per GPU, the same contiguous
ceil-chunking — rounded to whole image groups so that the questions of one image (POPE: 6) stay
on one rank and share its ViT features and prompt-prefix KV — weights replicated, no data-path
collective, and ONE gather of the per-question results at the end of the shard (RCCL over xGMI on GPUs,
gloo in the CPU tests): `ShardPlan` + `gather_results`, used by bench.py --strong and by the eval drivers (pope_driver.run_pope,
mme_driver.run_mme, blip_driver.run_blip_pope: each takes rank / world or reads the initialised process group, runs its chunk, and
rank 0 writes the answers file and scores it - the reference instead leaves one JSONL per process to be concatenated by its shell
scripts, scripts/pope/run_dataset.sh:14-33)."""
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


def rank_world(rank=None, world=None):
    """(rank, world) of this process: the arguments if given, else the initialised process group's, else (0, 1)."""
    if rank is not None and world is not None:
        return int(rank), int(world)
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class ShardPlan:
    """The split of one evaluation run over `world` ranks, computed identically on every rank: `keys[p]` names the group of position p
    (the image of the p-th question in the driver's image-sorted order); a group = a run of equal consecutive keys and never straddles
    two ranks (its questions share ViT features and prompt-prefix KV); rank k takes the k-th contiguous ceil-chunk of GROUPS - the
    reference's get_chunk (MME/run_llava.py:32-40) in units of images.  `mine`: this rank's positions; `capacity`: the largest chunk
    (the row count of every rank's block in the one result gather - known everywhere without an exchange)."""

    def __init__(self, keys: Sequence, rank=None, world=None):
        self.rank, self.world = rank_world(rank, world)
        self.n_total = len(keys)
        starts = [p for p in range(self.n_total) if p == 0 or keys[p] != keys[p - 1]] + [self.n_total]
        n_groups = len(starts) - 1
        per = math.ceil(n_groups / self.world) if n_groups else 0
        self.chunks = []
        for k in range(self.world):
            g0, g1 = min(n_groups, k * per), min(n_groups, (k + 1) * per)
            self.chunks.append(range(starts[g0], starts[g1]))
        self.mine = self.chunks[self.rank]
        self.capacity = max([len(c) for c in self.chunks] + [1])


def gather_results(local_ids: torch.Tensor, tokens: torch.Tensor, n_tokens: torch.Tensor, top_tok: torch.Tensor | None,
                   top_prob: torch.Tensor | None, n_total: int, pad: int = 0, capacity: int | None = None, width: int | None = None,
                   world: int | None = None) -> dict:
    """The per-question payload of SURVEY.md section 8(e) - {qid, n_tokens, tokens[T], top-n tokens, top-n probabilities} - in ONE
    collective: every rank contributes a fixed-shape int64 block [capacity, 2 + width + 2 k] (the fp32 probabilities bit-cast into
    it; unused rows carry qid -1) to one `all_gather_into_tensor`, a few hundred bytes per question.  `capacity` (rows per rank:
    ShardPlan.capacity) and `width` (token columns: max_new_tokens) are known on every rank from the shard plan, so there is no
    size exchange; callers that cannot know them (ragged ad-hoc use) leave them None and pay one extra all_reduce(MAX).
    `world` = 1: the caller's plan has ONE rank (a driver run unsharded inside a larger job): nothing is exchanged.
    Returns, on every rank, tensors ordered by question index; `count[q]` = how many ranks delivered question q (a partition: all 1)."""
    dev = tokens.device
    n_local, T = tokens.shape
    k = 0 if top_tok is None else int(top_tok.shape[1])
    import os
    distributed = (dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("VDD_FORCE_DIST") == "1")
                   and (world is None or world > 1 or os.environ.get("VDD_FORCE_DIST") == "1"))
    if distributed and world is not None and world != dist.get_world_size():
        raise ValueError(f"gather_results: the shard plan has {world} ranks, the process group {dist.get_world_size()}")
    if distributed and (capacity is None or width is None):
        m = torch.tensor([n_local, T], dtype=torch.long, device=dev)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        capacity, width = max(int(m[0].item()), 1), int(m[1].item())
    if not distributed:
        capacity, width = max(n_local, 1), T if width is None else max(width, T)
    if n_local > capacity or T > width:
        raise ValueError(f"gather_results: {n_local} x {T} local results do not fit the declared block {capacity} x {width}")
    cols = 2 + width + 2 * k
    buf = torch.full((capacity, cols), -1, dtype=torch.long, device=dev)                 # qid -1 = padding row
    buf[:, 2: 2 + width] = pad
    buf[:n_local, 0], buf[:n_local, 1] = local_ids.long(), n_tokens.long()
    buf[:n_local, 2: 2 + T] = tokens.long()
    if k:
        buf[:n_local, 2 + width: 2 + width + k] = top_tok.long()
        buf[:n_local, 2 + width + k:] = top_prob.float().contiguous().view(torch.int32).long()
    if distributed:
        world = dist.get_world_size()
        allb = torch.empty((world * capacity, cols), dtype=torch.long, device=dev)
        try:
            dist.all_gather_into_tensor(allb, buf)
        except (RuntimeError, NotImplementedError):                                      # a backend without the flat form: same data, list form
            dist.all_gather(list(allb.view(world, capacity, cols).unbind(0)), buf)
    else:
        allb = buf
    ok = allb[:, 0] >= 0                                                                 # one vectorised unpack for all ranks' blocks
    rows, q = allb[ok], allb[ok, 0]
    out = {"tokens": torch.full((n_total, width), pad, dtype=torch.long, device=dev), "n_tokens": torch.zeros(n_total, dtype=torch.long, device=dev),
           "count": torch.zeros(n_total, dtype=torch.long, device=dev)}
    out["count"].index_add_(0, q, torch.ones_like(q))
    out["n_tokens"][q] = rows[:, 1]
    out["tokens"][q] = rows[:, 2: 2 + width]
    if k:
        out["top_tok"] = torch.full((n_total, k), -1, dtype=torch.long, device=dev)
        out["top_prob"] = torch.zeros(n_total, k, dtype=torch.float32, device=dev)
        out["top_tok"][q] = rows[:, 2 + width: 2 + width + k]
        out["top_prob"][q] = rows[:, 2 + width + k:].to(torch.int32).view(torch.float32)
    return out


def init_from_env(device_index: int | None = None):
    """Process-group set-up of a driver started by torchrun (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in the environment): backend
    "nccl" (= RCCL over xGMI) bound to this rank's GPU; VDD_DIST_BACKEND=gloo / VDD_FORCE_DEVICE=i for single-GPU test boxes.
    Returns (rank, world, device).  Without torchrun's variables: (0, 1, cuda:0) and no process group."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("VDD_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0"))) if device_index is None else device_index
    device = torch.device(f"cuda:{local}")
    if world > 1 and not (dist.is_available() and dist.is_initialized()):
        torch.cuda.set_device(device)
        backend = os.environ.get("VDD_DIST_BACKEND", "nccl")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": device} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, device


def resolve_batch_invariant(batch_invariant, world: int, generate_kw: dict) -> bool:
    """Do the drivers decode in batch-invariant mode (ops.GEMM_BATCH_INVARIANT: one arithmetic form per op, a row's results independent of
    its batch)?  An explicit True / False wins.  Default: yes for DETERMINISTIC decodes - cd_greedy, top_k = 1, do_sample = False - whose
    answers are then the same token for token on 1 and on N ranks and for any batch size ("top-k=1 runs are shard-invariant");
    no for sampled runs, whose per-rank seeds make the shards differ from a 1-rank run anyway (`world` is accepted for that statement only)."""
    if batch_invariant is not None:
        return bool(batch_invariant)
    return bool(generate_kw.get("cd_greedy") or generate_kw.get("top_k") == 1 or generate_kw.get("do_sample") is False)
