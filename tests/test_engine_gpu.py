"""The branch-batched engine vs a plain-PyTorch fp32 reference LLaVA driven by the oracle loop
(i.e. by the reference's decoding semantics), on a tiny random model that exercises every kernel
path (ViT, projector, splice, prefix-shared prefill, ragged decode, fused sampling tail)."""
import numpy as np
import pytest
import torch

from oracle import vdd_oracle as O
from ref_llava import RefLlava

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", params=[torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def eng(request):
    """Every engine test runs on a bf16 engine (BASELINE config #2) and on an fp16 engine (the dtype of the reference's drivers,
    builder.py:40); the fp32 reference rounds its inputs through the same storage type (ref_llava.RefLlava.store)."""
    from llava_align_amd.engine import LlavaWeights, VddLlavaEngine, preset
    cfg = preset("tiny")
    w = LlavaWeights.random(cfg, DEV, seed=3, std=0.06, dtype=request.param)
    e = VddLlavaEngine(cfg, weights=w, device=DEV, t_max=256, use_graph=False)
    assert e.dtype == request.param
    return e


@pytest.fixture(scope="module")
def ref(eng):
    return RefLlava(eng.w, device=DEV)


def prompts(n_img=2, per_img=3, seed=0, vocab=1000, n_sys=12, n_txt=(5, 9)):
    rng = np.random.default_rng(seed)
    sys_tok = [1] + rng.integers(3, vocab, size=n_sys - 1).tolist()
    ids, imgs, base = [], [], []
    for i in range(n_img):
        im = torch.randn(3, 56, 56, generator=torch.Generator().manual_seed(100 + i))
        for _ in range(per_img):
            txt = rng.integers(3, vocab, size=int(rng.integers(*n_txt))).tolist()
            ids.append(torch.tensor(sys_tok + [-200] + txt))
            imgs.append(im)
    return ids, imgs


def cos(a, b):
    return torch.nn.functional.cosine_similarity(a.float().flatten(), b.float().flatten(), dim=0).item()


def test_vit_and_projector_match_fp32_reference(eng, ref):
    ims = torch.stack([torch.randn(3, 56, 56, generator=torch.Generator().manual_seed(i)) for i in range(3)])
    got = eng.vit(ims).float()
    want = ref.encode_images(ims)
    assert cos(got, want) > 0.9995
    assert (got - want).abs().max().item() <= 0.05 * want.abs().max().item()          # bf16 activations through 2 ViT layers


def test_step0_logits_per_branch(eng, ref):
    ids, imgs = prompts()
    for share in (True, False):
        eng.generate(ids, images=imgs, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, max_new_tokens=1, cd_greedy=True,
                     use_dd=True, use_dd_unk=True, share_prefix=share)
        L, Q = eng.debug_logits0.float().cpu(), len(ids)
        for q in range(Q):
            i = ids[q]
            unk = i.clone(); unk[unk == -200] = 0                              # vcd_sample.py:154-155
            want = [ref(input_ids=i[None], images=imgs[q][None]).logits[0, -1],
                    ref(input_ids=unk[None], images=None).logits[0, -1],
                    ref(input_ids=i[i != -200][None], images=None).logits[0, -1]]   # :160
            for b in range(3):
                w_ = want[b].float()
                assert (L[b * Q + q] - w_).abs().max().item() <= 0.03 * w_.abs().max().item() + 0.02, (share, q, b)


def run_ref(ref, ids, img, mode_kw, n_new, img_cd=None, eos=None, pad=None):
    kw = dict(images=img[None], attention_mask=torch.ones(1, ids.numel(), dtype=torch.long), use_cache=True,
              cd_alpha=1.0, cd_beta=0.1, **mode_kw)
    if img_cd is not None:
        kw["images_cd"] = img_cd[None]
    ref.calls.clear()
    return O.reference_loop(ref, ids[None].clone(), warp=O.WarpConfig(temperature=0.5), max_length=ids.numel() + n_new,
                            pad_token_id=pad, eos_token_id=eos, pick=O.pick_argmax, **kw)


MODES = {"plain": {}, "dd_unk": {"use_dd_unk": True}, "dd": {"use_dd": True}, "both": {"use_dd": True, "use_dd_unk": True}}


@pytest.mark.both_scalar_forms
@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("share", [True, False])
def test_generation_matches_reference_semantics(eng, ref, mode, share):
    ids, imgs = prompts()
    n_new = 6
    out = eng.generate(ids, images=imgs, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, max_new_tokens=n_new, cd_greedy=True,
                       output_scores=True, share_prefix=share, **MODES[mode])
    # bf16 engine vs fp32 reference: raw logits agree to ~0.06 (checked per branch in test_step0_logits_per_branch);
    # the contrast amplifies that by ((1+a) + a) / T = 6x, so post-warp scores are compared at 0.4 and tokens only
    # where the reference's top-1 margin is clear of that noise.
    tol = 0.4 if mode != "plain" else 0.15
    checked = 0
    for q in range(len(ids)):
        r = run_ref(ref, ids[q], imgs[q], MODES[mode], n_new)
        want = r.sequences[0, ids[q].numel():].tolist()
        got = out.tokens[q].tolist()
        for step in range(n_new):
            s_got, s_want = out.scores[step][q].float().cpu(), r.scores[step][0].float().cpu()
            fin = torch.isfinite(s_got) & torch.isfinite(s_want)
            # tokens whose main-branch logit sits within bf16 noise of the plausibility cutoff may fall on either side
            assert fin.sum() >= 1 and (torch.isfinite(s_got) ^ torch.isfinite(s_want)).sum() <= 3 + 0.05 * int(fin.sum())
            assert (s_got[fin] - s_want[fin]).abs().max().item() <= tol, (q, step)
            top2 = torch.topk(s_want, 2).values
            if (top2[0] - top2[1]).item() > 2 * tol:
                assert got[step] == want[step], (q, step)
                checked += 1
            if got[step] != want[step]:
                break                      # near-tie flipped: the continuations are different sequences from here on
    assert checked >= len(ids)             # the comparison actually bit on a fair number of tokens
    if share and mode != "plain":
        assert out.stats["prefill_tokens"] < out.stats["unshared_prefill_tokens"]


@pytest.fixture
def batch_invariant():
    """Data-parallel GEMM schedule only: a row's projections are then accumulated in one fixed order whatever else is in the
    batch, so two differently batched runs can be compared token for token (ops.GEMM_BATCH_INVARIANT)."""
    from llava_align_amd import ops
    old, ops.GEMM_BATCH_INVARIANT = ops.GEMM_BATCH_INVARIANT, True
    yield
    ops.GEMM_BATCH_INVARIANT = old


def test_prefix_sharing_is_transparent(eng, batch_invariant):
    ids, imgs = prompts(seed=5)
    kw = dict(images=imgs, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, max_new_tokens=5, cd_greedy=True, output_scores=True,
              use_dd=True, use_dd_unk=True)
    a = eng.generate(ids, share_prefix=True, **kw)
    b = eng.generate(ids, share_prefix=False, **kw)
    tol = 0.2                                  # grouped path rounds P to bf16 for the MFMA; x6 contrast gain
    agree = 0
    for q in range(len(ids)):                  # attention over [prefix | own] tiles its softmax differently from one segment, so
        for step, (sa, sb) in enumerate(zip(a.scores, b.scores)):     # the two runs agree to rounding, not bit for bit
            fin = torch.isfinite(sa[q]) & torch.isfinite(sb[q])
            assert (sa[q][fin].float() - sb[q][fin].float()).abs().max().item() <= tol, (q, step)
            top2 = torch.topk(sa[q].float(), 2).values
            if (top2[0] - top2[1]).item() > 2 * tol:
                assert a.tokens[q, step] == b.tokens[q, step], (q, step)
            if a.tokens[q, step] != b.tokens[q, step]:
                break                          # a near-tie went the other way: different continuations from here on
            agree += 1
    assert agree >= 4 * len(ids)               # and they do agree on almost every token


def test_vcd_branch_only_counts_at_step_zero(eng, ref):
    """from step 1 on the noisy-image branch runs on the main cache, c == v."""
    ids, imgs = prompts(n_img=1, per_img=2, seed=7)
    noisy = [im * 0.3 + torch.randn(3, 56, 56, generator=torch.Generator().manual_seed(55)) for im in imgs]
    out = eng.generate(ids, images=imgs, images_cd=noisy, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, max_new_tokens=4,
                       cd_greedy=True, output_scores=True)
    for q in range(len(ids)):
        r = run_ref(ref, ids[q], imgs[q], {}, 4, img_cd=noisy[q])
        fin = torch.isfinite(out.scores[0][q].cpu()) & torch.isfinite(r.scores[0][0].cpu())
        assert (out.scores[0][q].float().cpu()[fin] - r.scores[0][0].float().cpu()[fin]).abs().max().item() <= 0.4
        top2 = torch.topk(r.scores[0][0].float(), 2).values
        if (top2[0] - top2[1]).item() > 0.8:
            assert out.tokens[q, 0].item() == r.sequences[0, ids[q].numel()].item()
    # steps >= 1 run with c == v: scores = fl(fl(2v) - v) / T on the survivors of the beta mask, i.e. the plain
    # logits' arg-max survives and wins -> same continuation as plain decoding from the same first token
    for q in range(len(ids)):
        for step in range(1, 4):
            sc = out.scores[step][q]
            assert int(torch.isfinite(sc).sum()) >= 1


def test_eos_pad_and_early_stop(eng):
    ids, imgs = prompts(seed=9)
    base = eng.generate(ids, images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, max_new_tokens=8, cd_greedy=True)
    eos = int(base.tokens[0, 2])                      # question 0 emits this at step 2
    out = eng.generate(ids, images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, max_new_tokens=8, cd_greedy=True,
                       eos_token_id=eos, pad_token_id=0, sync_every=1)
    for q in range(len(ids)):
        row, ref_row = out.tokens[q].tolist(), base.tokens[q].tolist()
        if eos in ref_row:
            k = ref_row.index(eos)
            assert row[:k + 1] == ref_row[:k + 1] and all(t == 0 for t in row[k + 1:])      # vcd_sample.py:260
        else:
            assert row == ref_row[:len(row)]
    with pytest.raises(ValueError, match="pad_token_id"):
        eng.generate(ids, images=imgs, eos_token_id=eos, max_new_tokens=2)


def test_sampled_mode_runs_and_is_seed_reproducible(eng):
    ids, imgs = prompts(seed=11)
    kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.5, temperature=1.0, top_p=0.9, max_new_tokens=6)
    a = eng.generate(ids, seed=1, **kw)
    b = eng.generate(ids, seed=1, **kw)
    c = eng.generate(ids, seed=2, **kw)
    assert torch.equal(a.tokens, b.tokens) and not torch.equal(a.tokens, c.tokens)


def test_hip_graph_replay_matches_eager(eng):
    from llava_align_amd.engine import VddLlavaEngine
    g = VddLlavaEngine(eng.cfg, weights=eng.w, device=DEV, t_max=256, use_graph=True)
    ids, imgs = prompts(seed=13)
    for kw in (dict(use_dd_unk=True, cd_greedy=True, temperature=0.5),
               dict(use_dd=True, use_dd_unk=True, temperature=1.0, top_p=0.9, seed=5),
               dict(use_dd_unk=True, temperature=1.0, seed=6, eos_token_id=[7, 11], pad_token_id=0)):
        a = eng.generate(ids, images=imgs, cd_alpha=1.0, cd_beta=0.2, max_new_tokens=7, output_scores=True, **kw)
        for rep in range(2):                               # second call replays the cached graph with fresh state
            b = g.generate(ids, images=imgs, cd_alpha=1.0, cd_beta=0.2, max_new_tokens=7, output_scores=True, **kw)
            assert b.stats["graph"] and not a.stats["graph"]
            assert torch.equal(a.tokens, b.tokens)
            assert all(torch.equal(x, y) for x, y in zip(a.scores, b.scores))


def test_grouped_prefix_attention_is_transparent(eng):
    """Grouped (MFMA prefix pass on the transposed-V copy) vs per-row attention: same scores up to the bf16 rounding
    of P inside the MFMA (x6 contrast gain), same tokens wherever the top-1 margin is clear of that."""
    ids, imgs = prompts(seed=17)
    kw = dict(images=imgs, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, max_new_tokens=6, cd_greedy=True, output_scores=True,
              use_dd=True, use_dd_unk=True)
    from llava_align_amd import ops
    fmax, ops.FUSED_ATTN_MAX_M = ops.FUSED_ATTN_MAX_M, 0        # (up to FUSED_ATTN_MAX_M rows the engine would not group at all)
    try:
        eng.group_attention = True
        a = eng.generate(ids, **kw)
        assert a.stats["n_groups"] > 0
        eng.group_attention = False
        b = eng.generate(ids, **kw)
    finally:
        eng.group_attention = True
        ops.FUSED_ATTN_MAX_M = fmax
    checked = 0
    for q in range(len(ids)):
        for step in range(6):
            sa, sb = a.scores[step][q].float(), b.scores[step][q].float()
            fin = torch.isfinite(sa) & torch.isfinite(sb)
            assert (sa[fin] - sb[fin]).abs().max().item() <= 0.3      # a few bf16 ulps at scores of ~10 (0.16 - 0.22 over prompt sets)
            top2 = torch.topk(sb, 2).values
            if (top2[0] - top2[1]).item() > 0.6:
                assert a.tokens[q, step] == b.tokens[q, step]
                checked += 1
            if a.tokens[q, step] != b.tokens[q, step]:
                break
    assert checked >= len(ids)


@pytest.mark.both_scalar_forms
def test_lavis_call_shape_inputs_embeds_with_vcd_embeddings(eng):
    """BASELINE config #5 call shape: the LM is driven with inputs_embeds (Q-Former output ++ text embeddings) and the
    noisy-image branch arrives as EMBEDDINGS in images_cd (blip2_vicuna_instruct.py:380-410, modeling_llama.py:764-792)."""
    from ref_llava import RefLavisLM
    ref = RefLavisLM(eng.w, device=DEV)
    g = torch.Generator().manual_seed(3)
    d = eng.cfg.lm.d
    embs = [torch.randn(32 + n, d, generator=g) * 0.3 for n in (9, 14, 11)]
    embs_cd = [e + torch.randn(e.shape, generator=g) * 0.2 for e in embs]
    n_new = 4
    out = eng.generate(None, inputs_embeds=embs, images_cd=embs_cd, cd_alpha=0.5, cd_beta=0.1, temperature=0.5, max_new_tokens=n_new,
                       cd_greedy=True, output_scores=True)
    plain = eng.generate(None, inputs_embeds=embs, temperature=0.5, max_new_tokens=n_new, cd_greedy=True, output_scores=True)
    assert out.tokens.shape == (3, n_new) and all(s.numel() == n_new for s in out.sequences)      # no prompt ids to echo
    checked = 0
    for q in range(3):
        for o, cd in ((out, embs_cd[q]), (plain, None)):
            kw = dict(inputs_embeds=embs[q][None], attention_mask=torch.ones(1, embs[q].shape[0], dtype=torch.long), use_cache=True,
                      cd_alpha=0.5, cd_beta=0.1)
            if cd is not None:
                kw["images_cd"] = cd[None]
            r = O.reference_loop(ref, torch.zeros(1, 0, dtype=torch.long), warp=O.WarpConfig(temperature=0.5), max_length=n_new,
                                 pad_token_id=None, eos_token_id=None, pick=O.pick_argmax, **kw)
            for step in range(n_new):
                s_got, s_want = o.scores[step][q].float().cpu(), r.scores[step][0].float().cpu()
                fin = torch.isfinite(s_got) & torch.isfinite(s_want)
                assert fin.sum() >= 1 and (s_got[fin] - s_want[fin]).abs().max().item() <= 0.4
                top2 = torch.topk(s_want, 2).values
                if (top2[0] - top2[1]).item() > 0.8:
                    assert o.tokens[q, step].item() == r.sequences[0, step].item()
                    checked += 1
                if o.tokens[q, step].item() != r.sequences[0, step].item():
                    break
    assert checked >= 4
    # use_dd_unk on slot-free embeddings = the reference's Qwen case: the image-free branch re-runs the same inputs
    dd = eng.generate(None, inputs_embeds=embs, use_dd_unk=True, max_new_tokens=2, cd_greedy=True)
    assert dd.stats["n_rows"] == 6 and dd.tokens.shape == (3, 2)


def test_qwen_shaped_lm_bias_and_large_vocab():
    """BASELINE config #4 LM shape: qkv bias (modeling_qwen.py:224-226), V=151936 (a row larger than LDS: the sampling kernel's
    workspace path inside the captured decode graph).  Image slots arrive as embeddings; quirk #4 (the cd branch re-runs the
    same inputs, c == v) is the call below with images_cd = inputs_embeds; the calibration passes are text-only ids."""
    from ref_llava import RefLavisLM
    from llava_align_amd.engine import LlavaWeights, VddLlavaEngine, preset
    cfg = preset("tiny-qwen")
    w = LlavaWeights.random(cfg, DEV, seed=5, std=0.06)
    e = VddLlavaEngine(cfg, weights=w, device=DEV)
    ref = RefLavisLM(w, device=DEV)
    g = torch.Generator().manual_seed(11)
    embs = [torch.randn(n, cfg.lm.d, generator=g) * 0.3 for n in (20, 27)]
    n_new = 3
    out = e.generate(None, inputs_embeds=embs, images_cd=embs, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, max_new_tokens=n_new,
                     cd_greedy=True, output_scores=True, n_top=10)
    noscore = e.generate(None, inputs_embeds=embs, images_cd=embs, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, max_new_tokens=n_new,
                         cd_greedy=True, output_scores=False)
    assert torch.equal(out.tokens, noscore.tokens)                       # workspace path == scores path
    checked = 0
    for q in range(2):
        r = O.reference_loop(ref, torch.zeros(1, 0, dtype=torch.long), warp=O.WarpConfig(temperature=0.5), max_length=n_new,
                             pad_token_id=None, eos_token_id=None, pick=O.pick_argmax, inputs_embeds=embs[q][None],
                             images_cd=embs[q][None], attention_mask=torch.ones(1, embs[q].shape[0], dtype=torch.long),
                             use_cache=True, cd_alpha=1.0, cd_beta=0.1)
        for step in range(n_new):
            s_got, s_want = out.scores[step][q].float().cpu(), r.scores[step][0].float().cpu()
            fin = torch.isfinite(s_got) & torch.isfinite(s_want)
            assert fin.sum() >= 1 and (s_got[fin] - s_want[fin]).abs().max().item() <= 0.4
            tok = out.tokens[q, step].item()                 # 152K near-flat logits: the pick must be a near-maximiser of the reference row
            assert s_want[tok].item() >= s_want.max().item() - 0.8
            checked += 1
            if tok != r.sequences[0, step].item():
                break
    assert checked >= 2
    assert out.top_tok.shape == (2, 10) and out.top_tok[:, 0].tolist() == out.tokens[:, 0].tolist()
    # text-only calibration pass (MME/run_qwen.py:103-109): ids, no image
    ids = [torch.randint(3, cfg.lm.vocab, (15,), generator=g), torch.randint(3, cfg.lm.vocab, (18,), generator=g)]
    t = e.generate(ids, temperature=0.5, max_new_tokens=2, cd_greedy=True, output_scores=True)
    ref2 = RefLlava(w, device=DEV)
    for q in range(2):
        r = O.reference_loop(ref2, ids[q][None], warp=O.WarpConfig(temperature=0.5), max_length=ids[q].numel() + 2, pad_token_id=None,
                             eos_token_id=None, pick=O.pick_argmax, attention_mask=torch.ones(1, ids[q].numel(), dtype=torch.long),
                             use_cache=True)
        s_got, s_want = t.scores[0][q].float().cpu(), r.scores[0][0].float().cpu()
        assert (s_got - s_want).abs().max().item() <= 0.4


def test_vit_graph_replay_equals_eager_forward():
    """A full batch of VisionTower.GRAPH_BATCH images replays a captured HIP graph; smaller batches and the first call of a
    process run eagerly: same features either way, and replays see NEW images (static input buffer refreshed)."""
    from llava_align_amd.engine import LlavaWeights, VddLlavaEngine, preset
    cfg = preset("tiny")
    e = VddLlavaEngine(cfg, weights=LlavaWeights.random(cfg, DEV, seed=3, std=0.06), device=DEV, use_graph=True)
    nb = e.vit.GRAPH_BATCH
    g = torch.Generator().manual_seed(4)
    for rep in range(3):
        imgs = torch.randn(nb, 3, cfg.vision.image, cfg.vision.image, generator=g)
        got = e.vit(imgs)
        assert nb in e.vit._graphs
        want = e.vit._forward(imgs)
        assert got.shape == want.shape and torch.equal(got, want), rep
    small = torch.randn(3, 3, cfg.vision.image, cfg.vision.image, generator=g)
    assert torch.equal(e.vit(small), e.vit._forward(small)) and 3 not in e.vit._graphs
    one = torch.randn(1, 3, cfg.vision.image, cfg.vision.image, generator=g)      # a single question's image: graphed too
    assert torch.equal(e.vit(one), e.vit._forward(one)) and 1 in e.vit._graphs
    assert torch.equal(e.vit(one + 1), e.vit._forward(one + 1))
    # through generate(): 16 distinct images -> the graph path feeds the prefill
    ids = [torch.tensor([1, 5, 6, -200, 7 + i, 8]) for i in range(nb)]
    imgs = [torch.randn(3, cfg.vision.image, cfg.vision.image, generator=g) for _ in range(nb)]
    a = e.generate(ids, images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, max_new_tokens=3, cd_greedy=True)
    e2 = VddLlavaEngine(cfg, weights=e.w, device=DEV, use_graph=False)
    b = e2.generate(ids, images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, max_new_tokens=3, cd_greedy=True)
    assert torch.equal(a.tokens, b.tokens)


def test_host_images_through_the_pinned_staging_buffers_equal_device_images():
    """CPU image tensors (what the reference's drivers hand to generate()) are stacked into two pinned staging buffers and uploaded
    asynchronously, VIT_CHUNK at a time: 2 chunks + 8 images = three chunks, so the first buffer is reused while its upload may
    still be in flight.  Same features as the same images already on the device, repeatedly, and in fp16 as well as fp32.  The chunk
    size itself is transparent up to the GEMM schedule (a 16-image and a 64-image forward cut K differently: rounding-sized)."""
    from llava_align_amd.engine import LlavaWeights, VddLlavaEngine, preset
    cfg = preset("tiny")
    e = VddLlavaEngine(cfg, weights=LlavaWeights.random(cfg, DEV, seed=3, std=0.06), device=DEV, use_graph=True)
    assert e.VIT_CHUNK == 64
    g = torch.Generator().manual_seed(9)
    for chunk in (16, 64):
        e.VIT_CHUNK = chunk
        n = 2 * chunk + 8
        for dt in (torch.float32, torch.float16):
            for rep in range(2):
                host = [torch.randn(3, cfg.vision.image, cfg.vision.image, generator=g).to(dt) for _ in range(n)]
                dev = [im.to(DEV) for im in host]
                fh, fd = e.image_features(host), e.image_features(dev)
                assert all(torch.equal(a, b) for a, b in zip(fh, fd)), (chunk, dt, rep)
        assert e._pin[0].is_pinned() and e._pin[0].shape[0] == chunk and e._pin_key[0] == torch.float16
    e.VIT_CHUNK = 16
    f16 = e.image_features(dev)
    e.VIT_CHUNK = 64
    f64 = e.image_features(dev)
    err = max(float((a.float() - b.float()).abs().max()) for a, b in zip(f16, f64))
    ref = max(float(a.float().abs().max()) for a in f16)
    assert err <= 2e-2 * ref, (err, ref)


def test_captured_steps_survive_a_later_larger_batch():
    """Every captured decode step owns its split-KV partials buffer: replaying the graph of a small ungrouped batch after a
    larger batch has run (which would have re-allocated a shared workspace) must give the same tokens as before."""
    from llava_align_amd.engine import LlavaWeights, VddLlavaEngine, preset
    cfg = preset("tiny")
    e = VddLlavaEngine(cfg, weights=LlavaWeights.random(cfg, DEV, seed=3, std=0.06), device=DEV, use_graph=True)
    g = torch.Generator().manual_seed(8)

    def batch(n):
        ids = [torch.tensor([1, 5, 6, -200] + torch.randint(3, 900, (4 + i % 3,), generator=g).tolist()) for i in range(n)]
        imgs = [torch.randn(3, cfg.vision.image, cfg.vision.image, generator=g) for _ in range(n)]     # distinct images: ungrouped
        return ids, imgs
    kw = dict(use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, max_new_tokens=6, cd_greedy=True)
    ids_a, imgs_a = batch(9)          # 18 rows: split-KV attention (more than the fused small-M kernel takes)
    ids_b, imgs_b = batch(20)
    first = e.generate(ids_a, images=imgs_a, **kw)
    assert first.stats["graph"]
    e.generate(ids_b, images=imgs_b, **kw)
    again = e.generate(ids_a, images=imgs_a, **kw)
    assert torch.equal(first.tokens, again.tokens)
    eager = VddLlavaEngine(cfg, weights=e.w, device=DEV, use_graph=False).generate(ids_a, images=imgs_a, **kw)
    assert torch.equal(first.tokens, eager.tokens)


def test_soak_graph_engine_equals_eager_engine_over_mixed_calls():
    """A long-lived engine (cached graphs, cached KV pools, reused runners) must give what a fresh eager engine gives for every
    call of a mixed sequence: different batch sizes, image sharing patterns, decoding modes, EOS on/off."""
    import random
    from llava_align_amd.engine import LlavaWeights, VddLlavaEngine, preset
    cfg = preset("tiny")
    w = LlavaWeights.random(cfg, DEV, seed=3, std=0.06)
    eg = VddLlavaEngine(cfg, weights=w, device=DEV, use_graph=True)
    ee = VddLlavaEngine(cfg, weights=w, device=DEV, use_graph=False)
    rnd = random.Random(5)
    g = torch.Generator().manual_seed(5)
    pool = [torch.randn(3, cfg.vision.image, cfg.vision.image, generator=g) for _ in range(6)]
    modes = [dict(), dict(use_dd_unk=True), dict(use_dd=True), dict(use_dd=True, use_dd_unk=True), dict(images_cd="noise")]
    for call in range(14):
        Q = rnd.choice([1, 2, 3, 6, 7, 12, 18])
        share = rnd.choice([1, 2, 6])                                   # questions per image
        ids, imgs = [], []
        for q in range(Q):
            n = rnd.randint(2, 7)
            ids.append(torch.tensor([1, 5, 6, -200] + [rnd.randint(3, 900) for _ in range(n)]))
            imgs.append(pool[(q // share) % len(pool)])
        kw = dict(modes[call % len(modes)])
        if kw.get("images_cd") == "noise":
            kw["images_cd"] = [im + 0.5 for im in imgs]
        if kw.get("use_dd") and not kw.get("use_dd_unk"):
            pass                                                         # use_dd alone is B=1-only in the reference; the engine batches it
        eos = dict(eos_token_id=[rnd.randint(3, 900)], pad_token_id=0) if call % 3 == 0 else {}
        args = dict(images=imgs, cd_alpha=1.0, cd_beta=0.1, temperature=0.6, max_new_tokens=rnd.choice([2, 5, 9]), cd_greedy=(call % 2 == 0),
                    seed=100 + call, **kw, **eos)     # odd calls SAMPLE: same Philox (seed, step, row) stream in both engines
        a, b = eg.generate(ids, **args), ee.generate(ids, **args)
        assert torch.equal(a.tokens, b.tokens), (call, Q, share, kw.keys())


def test_generate_validates_prompt_ids_on_the_host(eng):
    """The embedding gather runs on the device inside a captured step: ids are checked before anything is launched."""
    ids, imgs = prompts(n_img=1, per_img=1)
    two_slots = torch.cat([ids[0], torch.tensor([-200, 5])])
    with pytest.raises(ValueError, match="image placeholders"):
        eng.generate([two_slots], images=imgs[:1], max_new_tokens=2)
    for bad in (-3, 1000, 10 ** 9):
        with pytest.raises(ValueError, match="outside"):
            eng.generate([torch.cat([ids[0], torch.tensor([bad])])], images=imgs[:1], max_new_tokens=2)


def test_embed_kernels_clamp_ids_instead_of_reading_wild_memory():
    from llava_align_amd import ops
    table = (torch.arange(50 * 8, device=DEV).view(50, 8) % 251).to(torch.bfloat16)
    ids = torch.tensor([3, -1, 49, 50, 1 << 40], device=DEV)
    got = ops.embed(ids, table)
    assert torch.equal(got, table[torch.tensor([3, 0, 49, 0, 0], device=DEV)])
    out = torch.zeros(5, 8, dtype=torch.bfloat16, device=DEV)
    ops.embed_scatter(torch.tensor([7, -200, 50, 2, 0], dtype=torch.int32, device=DEV), torch.arange(5, dtype=torch.int32, device=DEV), table, out)
    assert torch.equal(out, table[torch.tensor([7, 0, 0, 2, 0], device=DEV)])


def test_sampling_stream_advances_between_calls_and_follows_manual_seed(eng):
    """with seed=None every generate() draws fresh numbers (the reference's multinomial advances torch's generator) and
    torch.manual_seed() reproduces a run; an explicit seed is a pure function of the seed."""
    ids, imgs = prompts(seed=11)
    kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=1e-4, temperature=1.5, max_new_tokens=6)   # wide distributions
    torch.manual_seed(1234)
    a1 = eng.generate(ids, **kw).tokens
    a2 = eng.generate(ids, **kw).tokens
    torch.manual_seed(1234)
    b1 = eng.generate(ids, **kw).tokens
    b2 = eng.generate(ids, **kw).tokens
    assert torch.equal(a1, b1) and torch.equal(a2, b2)            # reproducible from the seed
    assert not torch.equal(a1, a2)                                 # but two calls do not share their random numbers
    s1, s2 = eng.generate(ids, seed=5, **kw).tokens, eng.generate(ids, seed=5, **kw).tokens
    assert torch.equal(s1, s2) and not torch.equal(s1, eng.generate(ids, seed=6, **kw).tokens)
    # the drop-in tail and the noise op draw from the same generator
    from llava_align_amd import contrast_sample, add_diffusion_noise
    v = torch.randn(8, 500, device=DEV)
    torch.manual_seed(7); t1 = contrast_sample(v).tokens.clone(); t2 = contrast_sample(v).tokens.clone()
    torch.manual_seed(7); u1 = contrast_sample(v).tokens.clone()
    assert torch.equal(t1, u1) and not torch.equal(t1, t2)
    img = torch.randn(3, 16, 16, device=DEV)
    torch.manual_seed(9); n1 = add_diffusion_noise(img, 500); n2 = add_diffusion_noise(img, 500)
    torch.manual_seed(9); m1 = add_diffusion_noise(img, 500)
    assert torch.equal(n1, m1) and not torch.equal(n1, n2)


@pytest.mark.parametrize("n_img,per_img", [(1, 1), (1, 2), (1, 3), (1, 5), (2, 4)])      # 2, 4, 6, 10, 16 rows: every row bucket of the prologue
def test_norm_fused_few_row_step_equals_the_unfused_step(eng, n_img, per_img):
    """<= 16 rows in flight: the decode step without RMSNorm launches (LanguageModel._decode_step_few_rows) against the 7-launch layer."""
    ids, imgs = prompts(n_img=n_img, per_img=per_img, seed=31)        # questions x 2 branches
    kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, max_new_tokens=8, cd_greedy=True, output_scores=True)
    assert eng.lm.fuse_norms
    from llava_align_amd import ops
    try:
        ops.FORCE_LAYER_FORM = "fused"     # the five-launch layer at EVERY row count up to 16 (left alone the engine takes whichever form it measured faster)
        eng._graphs.clear()
        a = eng.generate(ids, **kw)
        ops.FORCE_LAYER_FORM = "plain"
        eng._graphs.clear()
        b = eng.generate(ids, **kw)
    finally:
        ops.FORCE_LAYER_FORM = None
        eng._graphs.clear()
    for sa, sb in zip(a.scores, b.scores):
        fin = torch.isfinite(sa) & torch.isfinite(sb)
        assert (torch.isfinite(sa) ^ torch.isfinite(sb)).sum() <= 2 * len(ids) and (sa[fin].float() - sb[fin].float()).abs().max().item() <= 0.25
    assert (a.tokens == b.tokens).float().mean().item() >= 0.75


def test_embedding_prompts_share_declared_prefixes_transparently(eng, batch_invariant):
    """inputs_embeds prompts (Qwen-VL / LAVIS call shape): `embeds_prefix=(key, n)` - prompts about one image start with the same n
    rows - and the degenerate image-free branch (same tensor) share prompt K/V; results equal the unshared run."""
    d = eng.cfg.lm.d
    g = torch.Generator(device=DEV).manual_seed(8)
    img = [(torch.randn(20, d, device=DEV, generator=g) * 0.3).to(eng.dtype) for _ in range(2)]
    embs, keys = [], []
    for q in range(5):
        txt = (torch.randn(4 + q, d, device=DEV, generator=g) * 0.3).to(eng.dtype)
        embs.append(torch.cat([img[q % 2], txt], 0))
        keys.append((f"image{q % 2}", 20))
    kw = dict(inputs_embeds=embs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, max_new_tokens=5, cd_greedy=True, output_scores=True)
    a = eng.generate(None, embeds_prefix=keys, **kw)
    b = eng.generate(None, share_prefix=False, **kw)
    c = eng.generate(None, **kw)                                   # same-tensor sharing only (main / image-free branch)
    assert a.stats["prefill_tokens"] == 2 * 20 + 2 * sum(4 + q for q in range(5))        # 2 image prefixes + both branches' text rows
    assert c.stats["prefill_tokens"] == sum(24 + q for q in range(5)) + 5 and b.stats["prefill_tokens"] == 2 * sum(24 + q for q in range(5))
    for x in (a, c):
        for sx, sb in zip(x.scores, b.scores):
            fin = torch.isfinite(sx) & torch.isfinite(sb)
            assert (torch.isfinite(sx) ^ torch.isfinite(sb)).sum() <= 2 and (sx[fin].float() - sb[fin].float()).abs().max().item() <= 0.3
        assert (x.tokens == b.tokens).float().mean().item() >= 0.8
    with pytest.raises(ValueError, match="embeds_prefix"):
        eng.generate(None, inputs_embeds=embs, embeds_prefix=keys[:2], max_new_tokens=1)


def test_declared_prefixes_never_cover_the_noised_cd_branch(eng, batch_invariant):
    """embeds_prefix is a promise about the MAIN prompts.  The VCD branch of the Qwen / LAVIS call shapes arrives as images_cd
    EMBEDDINGS whose image rows carry fresh noise per question (run_qwen.py:182, blip_calibrate.py:80): two questions about one
    image must each be contrasted against THEIR noised rows (all cd rows of a key used to share the first
    question's prefix slot, silently)."""
    d = eng.cfg.lm.d
    g = torch.Generator(device=DEV).manual_seed(11)
    img = (torch.randn(20, d, device=DEV, generator=g) * 0.3).to(eng.dtype)
    embs, embs_cd, keys = [], [], []
    for q in range(4):
        txt = (torch.randn(5 + q, d, device=DEV, generator=g) * 0.3).to(eng.dtype)
        noisy = (img.float() + torch.randn(20, d, device=DEV, generator=g) * 0.5).to(eng.dtype)      # a different draw per question
        embs.append(torch.cat([img, txt], 0))
        embs_cd.append(torch.cat([noisy, txt], 0))
        keys.append(("image0", 20))
    kw = dict(inputs_embeds=embs, images_cd=embs_cd, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, max_new_tokens=3, cd_greedy=True, output_scores=True)
    a = eng.generate(None, embeds_prefix=keys, **kw)
    b = eng.generate(None, share_prefix=False, **kw)
    # the main rows share the image prefix (1 x 20 rows + texts); every cd row is prefilled whole (4 x (20 + text))
    assert a.stats["prefill_tokens"] == 20 + sum(5 + q for q in range(4)) + sum(25 + q for q in range(4))
    L = eng.debug_logits0.float()
    eng.generate(None, share_prefix=False, **{**kw, "max_new_tokens": 1})
    Lb = eng.debug_logits0.float()
    assert (L[4:] - Lb[4:]).abs().max().item() <= 0.06 * Lb[4:].abs().max().item()          # cd logits: each question against its own noise
    assert (Lb[4] - Lb[5]).abs().max().item() > 0.2 * Lb[4:].abs().max().item()             # ... which really differs between questions
    for sa, sb in zip(a.scores, b.scores):
        fin = torch.isfinite(sa) & torch.isfinite(sb)
        assert (torch.isfinite(sa) ^ torch.isfinite(sb)).sum() <= 2 and (sa[fin].float() - sb[fin].float()).abs().max().item() <= 0.3


def test_text_only_prompts_share_their_common_system_prompt(eng, batch_invariant):
    """The content-free prior passes of the calibrate drivers (llava_calibrate.py:46-61): no image slot; the conversation template's system
    prompt - the longest prefix common to every prompt - is prefilled once, results unchanged."""
    rng = np.random.default_rng(17)
    sys_tok = [1] + rng.integers(3, 1000, size=20).tolist()
    ids = [torch.tensor(sys_tok + rng.integers(3, 1000, size=int(rng.integers(4, 9))).tolist()) for _ in range(6)]
    kw = dict(temperature=0.5, max_new_tokens=4, cd_greedy=True, output_scores=True, n_top=10)
    a = eng.generate(ids, **kw)
    b = eng.generate(ids, share_prefix=False, **kw)
    assert a.stats["prefill_tokens"] == 21 + sum(i.numel() - 21 for i in ids) < b.stats["prefill_tokens"] == sum(i.numel() for i in ids)
    for sa, sb in zip(a.scores, b.scores):
        assert (sa.float() - sb.float()).abs().max().item() <= 0.15
    assert (a.tokens == b.tokens).float().mean().item() >= 0.8
    one = eng.generate(ids[:1], **kw)                                  # a single prompt: nothing to share
    assert one.stats["prefill_tokens"] == ids[0].numel()


def test_streamer_receives_prompt_then_every_token_then_end(eng):
    """`streamer=` (HF BaseStreamer protocol; the reference's loop: vcd_sample.py:264-265 put(next_tokens.cpu()), :299-300 end(); HF's
    generate() puts the prompt ids first): one question per call, tokens arrive one per step, the stream stops with the EOS token."""
    class Rec:
        def __init__(self):
            self.items, self.ended = [], 0
        def put(self, value):
            assert value.device.type == "cpu"
            self.items.append(value.clone())
        def end(self):
            self.ended += 1
    ids, imgs = prompts(n_img=1, per_img=1, seed=41)
    kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, max_new_tokens=12, seed=5)
    plain = eng.generate(ids, **kw)
    rec = Rec()
    out = eng.generate(ids, streamer=rec, **kw)
    assert torch.equal(out.tokens, plain.tokens) and rec.ended == 1
    assert rec.items[0].shape == (1, ids[0].numel()) and torch.equal(rec.items[0][0], ids[0])
    assert torch.equal(torch.cat([t.reshape(1) for t in rec.items[1:]]), out.tokens[0].cpu())
    # EOS: the stream ends with the EOS token, as the loop does (no pad tokens are streamed)
    eos = int(plain.tokens[0, 3])
    rec2 = Rec()
    o2 = eng.generate(ids, streamer=rec2, eos_token_id=eos, pad_token_id=0, **kw)
    got = torch.cat([t.reshape(1) for t in rec2.items[1:]])
    assert int(got[-1]) == eos and torch.equal(got, o2.tokens[0, :got.numel()].cpu()) and rec2.ended == 1
    assert eos not in got[:-1].tolist()
    ids2, imgs2 = prompts(n_img=1, per_img=2, seed=41)
    with pytest.raises(ValueError):
        eng.generate(ids2, images=imgs2, streamer=Rec(), max_new_tokens=2)


def test_output_attentions_for_one_question_is_the_last_layer_map_of_step_zero(eng, ref):
    """llava_calibrate.py:175,180-182: generate(output_attentions=True) -> ['attentions'][0][-1], [1, H, T, T] of the spliced prompt.
    The engine computes exactly that one map (ops.attention_probs over the prefill's q and cached K); against the fp32 reference."""
    ids, imgs = prompts(n_img=1, per_img=1, seed=29)
    for share in (True, False):
        out = eng.generate(ids, images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, max_new_tokens=3, cd_greedy=True,
                           output_attentions=True, share_prefix=share)
        m = out["attentions"][0][-1]
        T = ids[0].numel() - 1 + eng.cfg.vision.n_patches
        assert m.shape == (1, eng.cfg.lm.n_heads, T, T) and m.dtype == eng.dtype and len(out["attentions"]) == 3
        ref(input_ids=ids[0][None], images=imgs[0][None])
        want = ref.last_attn.float()
        assert (m.float() - want).abs().max().item() <= 0.02 and not m.float().triu(1).any()
        avg = torch.mean(m, dim=1).squeeze()                        # what the driver does with it (:182)
        assert avg.shape == (T, T)
        with pytest.raises(IndexError, match="LAST layer"):
            out["attentions"][0][0]
        with pytest.raises(IndexError, match="step 0"):
            out["attentions"][1]
    many_ids, many_imgs = prompts()
    o2 = eng.generate(many_ids, images=many_imgs, max_new_tokens=1, output_attentions=True)       # a batch: accepted, nothing materialised
    with pytest.raises(KeyError, match="attentions"):
        o2["attentions"]


@pytest.mark.parametrize("n_img,per_img", [(1, 1), (2, 4), (5, 6)])         # 2 rows (weight-streaming + fused norms), 16 rows, 60 rows (MFMA GEMM)
def test_vocabulary_that_is_not_a_multiple_of_eight(n_img, per_img):
    """resize_token_embeddings(len(tokenizer)) after add_tokens leaves V = 32001 ... 32003 (builder.py:127-132).  lm_head then
    runs on a zero-padded copy and the logits are a strided [rows, V] VIEW of the [rows, V_pad] product; that view flows through the fused
    tail, its scores buffer, the repetition-penalty kernel, top-n and the captured graph.  V = 1003 here, against the fp32 reference."""
    from dataclasses import replace
    from llava_align_amd.engine import LlavaWeights, VddLlavaEngine, preset
    cfg = preset("tiny")
    cfg = replace(cfg, lm=replace(cfg.lm, vocab=1003))
    w = LlavaWeights.random(cfg, DEV, seed=11, std=0.06)
    ids, imgs = prompts(n_img=n_img, per_img=per_img, seed=5, vocab=1003)
    kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, max_new_tokens=5, cd_greedy=True, output_scores=True, n_top=10)
    outs = {}
    for graph in (False, True):
        e = VddLlavaEngine(cfg, weights=w, device=DEV, t_max=256, use_graph=graph)
        assert e.lm.lm_head.shape[0] == 1008 and e.lm.lm_head.data_ptr() != w.t["lm_head"].data_ptr()
        outs[graph] = e.generate(ids, **kw)
        o = outs[graph]
        assert all(s.shape == (len(ids), 1003) for s in o.scores) and int(o.tokens.max()) < 1003 and int(o.top_tok.max()) < 1003
        # with a repetition penalty (the contrast-only / plain split materialises the scores row in between) and sampling
        # (HF's processor gathers scores[input_ids]: only slot-free prompts combine with it - the text-only prior calls of the drivers)
        txt = [i[i != -200] for i in ids]
        r = e.generate(txt, temperature=0.5, max_new_tokens=5, repetition_penalty=1.3, seed=4, output_scores=True)
        assert r.tokens.shape == o.tokens.shape and int(r.tokens.max()) < 1003 and r.scores[0].shape == (len(ids), 1003)
    assert torch.equal(outs[False].tokens, outs[True].tokens)
    for a, b in zip(outs[False].scores, outs[True].scores):
        assert torch.equal(a, b)
    ref = RefLlava(w, device=DEV)
    checked = 0
    for q in range(0, len(ids), max(1, len(ids) // 3)):
        r = run_ref(ref, ids[q], imgs[q], {"use_dd_unk": True}, 5)
        for step in range(5):
            s_got, s_want = outs[True].scores[step][q].float().cpu(), r.scores[step][0].float().cpu()
            fin = torch.isfinite(s_got) & torch.isfinite(s_want)
            assert fin.sum() >= 1 and (torch.isfinite(s_got) ^ torch.isfinite(s_want)).sum() <= 3 + 0.1 * int(fin.sum())
            assert (s_got[fin] - s_want[fin]).abs().max().item() <= 0.4, (q, step)
            top2 = torch.topk(s_want, 2).values
            if (top2[0] - top2[1]).item() > 0.8 and bool(fin[outs[True].tokens[q, step]]):
                assert outs[True].tokens[q, step].item() == r.sequences[0, ids[q].numel() + step].item(), (q, step)
                checked += 1
            if outs[True].tokens[q, step].item() != r.sequences[0, ids[q].numel() + step].item():
                break
    assert checked >= 2


def test_reuse_prefill_decodes_again_from_the_kept_state():
    """A sweep over sampling settings asks for the same prompts again (MME/run_llava.py:281-318 runs 51 settings over one question file): with
    reuse_prefill the second call skips the vision tower and the prefill and decodes from the first call's pools and step-0 logits - tokens and
    step-0 top-10 equal a fresh call's for the same seed, for another temperature too; any other call into the same pools drops the kept state."""
    from test_engine_shapes_gpu import _engine, _prompts
    eng = _engine(dict(d=4096, n_heads=32, n_kv_heads=32, head_dim=128, ffn=11008, vocab=32000), n_layers=2, vit_layers=2)
    ids, imgs = _prompts(4, 6, 32000, seed=3)
    other_ids, other_imgs = _prompts(2, 6, 32000, seed=4)
    kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, max_new_tokens=8, n_top=10, seed=5)
    towers = []
    real = eng.image_features
    eng.image_features = lambda *a, **k: towers.append(1) or real(*a, **k)
    same = lambda a, b: torch.equal(a.tokens, b.tokens) and torch.equal(a.top_tok, b.top_tok) and torch.equal(a.top_prob, b.top_prob)
    fresh = eng.generate(ids, temperature=0.7, **kw)
    first = eng.generate(ids, temperature=0.7, reuse_prefill=True, **kw)
    n0 = len(towers)
    again = eng.generate(ids, temperature=0.7, reuse_prefill=True, **kw)
    cold = eng.generate(ids, temperature=0.3, top_p=0.9, **kw)
    warm = eng.generate(ids, temperature=0.7, reuse_prefill=True, **kw)                       # `cold` overwrote the pools: a miss that prefills again
    n1 = len(towers)
    swept = eng.generate(ids, temperature=0.3, top_p=0.9, reuse_prefill=True, **kw)
    short = eng.generate(ids, temperature=0.3, top_p=0.9, reuse_prefill=True, **dict(kw, max_new_tokens=3))
    assert not first.stats["prefill_reused"] and again.stats["prefill_reused"] and again.stats["prefill_tokens"] == 0 and n0 + 2 == n1 - 0
    assert not warm.stats["prefill_reused"] and swept.stats["prefill_reused"] and short.stats["prefill_reused"] and len(towers) == n1
    assert same(fresh, first) and same(fresh, again) and same(fresh, warm) and same(cold, swept)
    assert torch.equal(short.tokens, cold.tokens[:, :3])
    assert not eng.generate(ids, temperature=0.3, reuse_prefill=True, **dict(kw, max_new_tokens=12)).stats["prefill_reused"]      # more new tokens than kept room
    eng.generate(other_ids, images=other_imgs, max_new_tokens=4, use_dd_unk=True)
    assert not eng.generate(ids, temperature=0.3, reuse_prefill=True, **dict(kw, max_new_tokens=12)).stats["prefill_reused"]
    text = [torch.tensor([t for t in r.tolist() if t != -200]) for r in ids]                   # text-only prompts (the prior passes of a sweep)
    a = eng.generate(text, images=None, max_new_tokens=1, n_top=10, temperature=0.5, reuse_prefill=True)
    b = eng.generate(text, images=None, max_new_tokens=1, n_top=10, temperature=0.9, reuse_prefill=True)
    c = eng.generate(text, images=None, max_new_tokens=1, n_top=10, temperature=0.9)
    assert b.stats["prefill_reused"] and torch.equal(b.top_prob, c.top_prob) and not torch.equal(a.top_prob, b.top_prob)
