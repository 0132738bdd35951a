"""Row retirement + growing own-KV pools (the reference's LLaVA-Bench setting: llava_sampling.py:100-116 - open-ended answers
of 20 - 1,000 tokens, max_new_tokens = 1024, B = 1 per call there; SURVEY 8e: "limited by load imbalance from variable output length").
The engine batches the questions; rows that emitted EOS leave the batch at the next host check and the own pools grow with the longest
live row instead of holding n_rows x (suffix + max_new_tokens) from the start."""
import numpy as np
import pytest
import torch

from test_engine_shapes_gpu import _engine, _prompts

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
W13B = dict(d=5120, n_heads=40, n_kv_heads=40, head_dim=128, ffn=13824, vocab=32000)


def _eos_set(n, seed, vocab=32000):
    return sorted(set(np.random.default_rng(seed).integers(3, vocab, size=n).tolist()))


def test_retired_run_produces_the_tokens_of_the_static_run():
    """13B widths (4 layers), use_dd + use_dd_unk = 3 branches, top-p 0.9, cd_greedy: 24 questions whose answers end anywhere between a
    few and 200 tokens (EOS = a random set of 160 ids: every step ends an answer with probability ~0.5 %), retire on vs off - token for
    token.  A row's arithmetic must not depend on who else is in the batch for that: the data-parallel GEMM schedule
    (ops.GEMM_BATCH_INVARIANT) and no retirement below 24 rows (below 9 the projections change to the weight-streaming kernels, whose
    K split differs); with the tuned schedules two STATIC runs of 24 and of 20 questions already part ways within a few tokens."""
    from llava_align_amd import ops
    old, ops.GEMM_BATCH_INVARIANT = ops.GEMM_BATCH_INVARIANT, True
    try:
        eng = _engine(W13B, n_layers=4, vit_layers=2)
        ids, imgs = _prompts(24, 1, 32000, seed=41)
        eos = _eos_set(160, 7)
        kw = dict(images=imgs, use_dd=True, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=1.0, top_p=0.9, max_new_tokens=200,
                  cd_greedy=True, eos_token_id=eos, pad_token_id=0, sync_every=8)
        eng.retire, eng.kv_chunk, eng.retire_min_rows = False, 32, 24
        a = eng.generate(ids, **kw)
        eng._kvs.clear(); eng._graphs.clear()
        eng.retire = True
        b = eng.generate(ids, **kw)
        eng.use_graph = False
        eng._kvs.clear(); eng._graphs.clear()
        c = eng.generate(ids, **kw)                                # the same without captured steps
    finally:
        ops.GEMM_BATCH_INVARIANT = old
    assert "retire_events" not in a.stats and b.stats["retire_events"] >= 4 and 24 <= b.stats["rows_at_end"] <= 48
    sizes = [r for _, r, _ in b.stats["retire_log"]]
    caps = [c_ for _, _, c_ in b.stats["retire_log"]]
    assert min(sizes) < 72 and caps == sorted(caps) and caps[0] < 100 and caps[-1] == 200      # the batch shrank, the pools grew
    eos_t = torch.tensor(eos, device=DEV)
    la = ((a.tokens[:, :, None] == eos_t).any(-1).float().argmax(1))
    assert int((la > 0).sum()) >= 12 and int(la.max() - la[la > 0].min()) >= 50          # the answers really have different lengths
    assert a.tokens.shape == b.tokens.shape and torch.equal(a.tokens, b.tokens) and torch.equal(b.tokens, c.tokens)
    for q in range(24):                                           # finished rows are padded as the reference pads them (vcd_sample.py:260)
        if la[q] > 0 and la[q] + 1 < b.tokens.shape[1]:
            assert bool((b.tokens[q, la[q] + 1:] == 0).all())


def test_llava_bench_setting_fits_one_gpu_with_retirement():
    """The reference's own LLaVA-Bench setting on ONE device: 13B at full depth (40 layers), 90 questions x 3 branches, top-p 0.9 sampling,
    max_new_tokens = 1024.  The static form needs 270 rows x 1,054 own tokens x 0.82 MB = 233 GB of own KV alone; with retirement and
    growing pools the whole call stays under 200 GB."""
    from llava_align_amd.engine import LlavaConfig, LlavaWeights, LMConfig, VddLlavaEngine, VisionConfig
    cfg = LlavaConfig(LMConfig(n_layers=40, max_pos=2048, **W13B), VisionConfig(layers=2), "llava-bench-13b-shape")
    eng = VddLlavaEngine(cfg, weights=LlavaWeights.random(cfg, DEV, seed=3, std=0.02, lm_head_gain=2.0), device=DEV, use_graph=True)
    ids, imgs = _prompts(90, 1, 32000, seed=43)
    eos = _eos_set(100, 9)                                       # geometric answer lengths, mean ~ 300 tokens, tail to 1,024
    torch.cuda.reset_peak_memory_stats()
    out = eng.generate(ids, images=imgs, use_dd=True, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=1.0, top_p=0.9, max_new_tokens=1024,
                       eos_token_id=eos, pad_token_id=0, seed=5, sync_every=8)
    peak = torch.cuda.max_memory_allocated() / 1e9
    eos_t = torch.tensor(eos, device=DEV)
    is_eos = (out.tokens[:, :, None] == eos_t).any(-1)
    lens = torch.where(is_eos.any(1), is_eos.float().argmax(1) + 1, torch.full((90,), out.tokens.shape[1], device=DEV))
    assert out.stats["n_rows"] == 270 and out.stats["retire_events"] >= 4 and out.stats["rows_at_end"] <= 90
    assert peak < 200.0, peak
    assert int(lens.min()) < 100 and int(lens.max()) > 500 and (bool(is_eos.any(1).all()) or out.tokens.shape[1] == 1024)
    for q in range(90):                                          # pad behind every answer
        assert bool((out.tokens[q, int(lens[q]):] == 0).all())
    print(f"peak {peak:.1f} GB, {out.stats}")
