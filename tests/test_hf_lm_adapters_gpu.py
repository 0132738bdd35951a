"""attach_blip_engine / attach_lm_engine / attach_qwen_engine (VERDICT r4 missing #2): the native front-end + engine behind the calls the
reference's InstructBLIP and Qwen-VL drivers make -
  experiments/eval/calibrate/blip_calibrate.py:84-87  -> Blip2VicunaInstruct.generate (blip2_vicuna_instruct.py:233-418)
  experiments/eval/MME/run_qwen.py:190-213            -> QWenLMHeadModel.generate (modeling_qwen.py:1044-1087)
on module-shaped test doubles (tests/hf_doubles.py), against the reference's semantics: tests/ref_blip.py (pinned to LAVIS' own modules by
tests/golden/blip_vectors.npz) for the towers, the fp32 LM of tests/ref_llava.py driven by the oracle loop for the language model."""
import pytest
import torch

import hf_doubles
import ref_blip
from oracle import vdd_oracle as O
from ref_llava import RefLavisLM

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _check_rows(got_scores, got_tokens, runs, n_new, tol=0.4):
    checked = 0
    for q, r in enumerate(runs):
        for step in range(min(n_new, len(r.scores))):
            s_got, s_want = got_scores[step][q].float().cpu(), r.scores[step][0].float().cpu()
            fin = torch.isfinite(s_got) & torch.isfinite(s_want)
            assert fin.sum() >= 1 and (torch.isfinite(s_got) ^ torch.isfinite(s_want)).sum() <= 3 + 0.05 * int(fin.sum()), (q, step)
            assert (s_got[fin] - s_want[fin]).abs().max().item() <= tol, (q, step)
            top2 = torch.topk(s_want, 2).values
            same = int(got_tokens[q, step]) == int(r.sequences[0, step])
            if fin.sum() > 1 and (top2[0] - top2[1]).item() > 2 * tol:
                assert same, (q, step)
                checked += 1
            if not same:
                break
    return checked


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_blip_generate_through_the_adapter(dtype):
    """The driver's main call with the VCD branch (blip_calibrate.py:80-87) and a prior call (plain sampling on another image, :94-98)
    through `attach_blip_engine`, (output_text, scores) as the driver unpacks them."""
    from llava_align_amd.hf_adapter import attach_blip_engine, detach_engine
    from llava_align_amd.vcd_add_noise import add_diffusion_noise
    model = hf_doubles.build_blip(DEV, dtype)
    eng, front = attach_blip_engine(model)
    assert eng.dtype == dtype and front.cfg.vit.head_dim == 88 and front.cfg.qf.cross_freq == 2 and front.cfg.qf.n_query == 8
    g = torch.Generator().manual_seed(5)
    imgs = torch.randn(2, 3, 56, 56, generator=g).to(dtype).float()
    prompts = ["Is there a dog in the image? Please answer this question with one word.", "Is the sky blue?"]
    torch.manual_seed(3)
    imgs_cd = torch.stack([add_diffusion_noise(im, 500) for im in imgs.to(DEV)])
    n_new = 5
    text, scores = model.generate({"image": imgs.to(DEV), "prompt": prompts}, use_nucleus_sampling=True, num_beams=1, top_p=1.0,
                                  repetition_penalty=1, images_cd=imgs_cd, cd_beta=0.1, max_length=n_new, cd_greedy=True)
    assert len(text) == 2 and scores.shape == (2, 1000) and all(isinstance(t, str) for t in text)
    # the reference's path on the same weights: LAVIS towers (fp32 restatement pinned by the goldens) -> inputs_embeds(_cd) -> patched sample()
    sd32 = {k: v.float().cpu() for k, v in model._lavis.items()}
    tq = model.tokenizer(prompts, truncation=True, max_length=model.max_txt_len)
    qf_ids = [r[m.bool()].tolist() for r, m in zip(tq.input_ids, tq.attention_mask)]
    tl = model.llm_tokenizer(prompts)
    llm_ids = [r[m.bool()].tolist() for r, m in zip(tl.input_ids, tl.attention_mask)]
    table = eng.w.t["embed"].float().cpu()
    ref = RefLavisLM(eng.w, device=DEV)
    rnd = lambda t: t.to(dtype).float()                                 # the engine stores embeddings in the model dtype
    runs, raw = [], None
    for q in range(2):
        e = torch.cat([rnd(ref_blip.inputs_llm(sd32, front.cfg, imgs[q:q + 1], [qf_ids[q]])[0]), table[llm_ids[q]]], 0)
        e_cd = torch.cat([rnd(ref_blip.inputs_llm(sd32, front.cfg, imgs_cd[q:q + 1].float().cpu(), [qf_ids[q]])[0]), table[llm_ids[q]]], 0)
        kw = dict(inputs_embeds=e[None], images_cd=e_cd[None], attention_mask=torch.ones(1, e.shape[0], dtype=torch.long), use_cache=True,
                  cd_beta=0.1)                                          # cd_alpha is NOT forwarded by the driver: the sampler's 0.5
        runs.append(O.reference_loop(ref, torch.zeros(1, 0, dtype=torch.long), warp=O.WarpConfig(temperature=1.0, top_k=50, top_p=1.0),
                                     max_length=n_new, pad_token_id=0, eos_token_id=2, pick=O.pick_argmax,
                                     processors=O.ProcessorList([O.MinLength(1, [2])]), **kw))
    out = eng.generate(None, inputs_embeds=front.build(imgs.to(DEV), llm_ids, eng.w.t["embed"], qf_ids, imgs_cd)[0],
                       images_cd=front.build(imgs.to(DEV), llm_ids, eng.w.t["embed"], qf_ids, imgs_cd)[1], do_sample=True, top_p=1.0, top_k=50,
                       temperature=1, max_length=n_new, min_length=1, repetition_penalty=1, cd_beta=0.1, eos_token_id=2, pad_token_id=0,
                       output_scores=True, cd_greedy=True)
    assert torch.equal(out.scores[0], scores)                            # the adapter made exactly this engine call
    assert _check_rows(out.scores, out.tokens, runs, n_new) >= 2
    mapped = out.tokens.clone(); mapped[mapped == 0] = 2                  # blip2_vicuna_instruct.py:414
    assert text == [t.strip() for t in model.llm_tokenizer.batch_decode(mapped)]
    # a prior pass: plain sampling on the zero image (blip_calibrate.py:96-98); beam search is not on the patched path
    t2, s2 = model.generate({"image": torch.zeros_like(imgs[:1]).to(DEV), "prompt": prompts[0]}, use_nucleus_sampling=True, num_beams=1,
                            top_p=1.0, repetition_penalty=1, max_length=1)
    assert len(t2) == 1 and s2.shape == (1, 1000)
    with pytest.raises(ValueError, match="num_beams"):
        model.generate({"image": imgs.to(DEV), "prompt": prompts})          # LAVIS' default num_beams = 5
    detach_engine(model)
    assert "generate" not in model.__dict__ and not hasattr(model, "_vdd_front")


def test_lavis_llm_generate_through_attach_lm_engine():
    """`self.llm_model.generate(inputs_embeds=[B, T, d] left-padded, attention_mask, images_cd=inputs_embeds_cd, ...)`
    (blip2_vicuna_instruct.py:390-410) on an HF Llama through `attach_lm_engine`: the reference's own tower code can stay in front."""
    from llava_align_amd.hf_adapter import attach_lm_engine, detach_engine
    model = hf_doubles.build_blip(DEV, torch.float16)
    lm = model.llm_model
    eng = attach_lm_engine(lm)
    g = torch.Generator().manual_seed(9)
    d, lens = 256, (14, 11)
    T = max(lens)
    emb, mask = torch.zeros(2, T, d), torch.zeros(2, T, dtype=torch.long)
    for i, n in enumerate(lens):
        emb[i, T - n:], mask[i, T - n:] = torch.randn(n, d, generator=g) * 0.4, 1
    emb_cd = emb + torch.randn(emb.shape, generator=g) * 0.2 * mask[..., None]
    emb, emb_cd = emb.half(), emb_cd.half()
    n_new = 4
    out = lm.generate(inputs_embeds=emb.to(DEV), attention_mask=mask.to(DEV), do_sample=True, top_p=0.9, temperature=1, num_beams=1,
                      max_length=n_new, min_length=1, repetition_penalty=1.0, length_penalty=1, num_return_sequences=1, images_cd=emb_cd.to(DEV),
                      cd_beta=0.1, cd_alpha=None, use_dd=None, use_dd_unk=None, return_dict_in_generate=True, output_scores=True, cd_greedy=True)
    assert out["sequences"].shape == (2, n_new) and len(out["scores"]) == n_new and out["scores"][0].shape == (2, 1000)
    ref = RefLavisLM(eng.w, device=DEV)
    runs = []
    for i, n in enumerate(lens):
        kw = dict(inputs_embeds=emb[i, T - n:][None].float(), images_cd=emb_cd[i, T - n:][None].float(),
                  attention_mask=torch.ones(1, n, dtype=torch.long), use_cache=True, cd_beta=0.1)
        runs.append(O.reference_loop(ref, torch.zeros(1, 0, dtype=torch.long), warp=O.WarpConfig(temperature=1.0, top_k=50, top_p=0.9),
                                     max_length=n_new, pad_token_id=0, eos_token_id=2, pick=O.pick_argmax,
                                     processors=O.ProcessorList([O.MinLength(1, [2])]), **kw))
    assert _check_rows(out["scores"], out["tokens"], runs, n_new) >= 2
    detach_engine(lm)


def test_qwen_generate_through_the_adapter():
    """run_qwen.py:190-213 verbatim keywords on a QWenLMHeadModel-shaped object: the caller's `transformer.visual` fills the <img> span
    (modeling_qwen.py:545-575), the LM (fused c_attn with bias, w2 = gate / w1 = up, plain rotary below seq_length) and the VDD passes are
    native; use_dd_unk re-runs the same inputs, images_cd goes through the visual tower again."""
    from llava_align_amd.hf_adapter import attach_qwen_engine, detach_engine, qwen_spliced_embeddings
    model = hf_doubles.build_qwen(DEV, torch.bfloat16)
    eng = attach_qwen_engine(model)
    assert eng.cfg.lm.qkv_bias and eng.cfg.lm.ffn == 512 and eng.cfg.lm.max_pos == 512
    V, eod = model.config.vocab_size, model.generation_config.eos_token_id
    txt = hf_doubles.word_ids("Is there a cat in the picture? Answer:", V - 40)
    ids = torch.tensor([model.image_prompt(txt)], device=DEV)
    img = torch.randn(1, 3, 16, 16, generator=torch.Generator().manual_seed(4)).to(DEV)
    img_cd = img + 0.7 * torch.randn(img.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
    # the splice: rows between <img> and </img> are the visual tower's, everything else the token embeddings
    e = qwen_spliced_embeddings(model, ids, img)[0]
    want = model.transformer.wte(ids)[0].clone()
    want[1:1 + model.img_rows] = model.transformer.visual(img)[0].to(want.dtype)
    assert torch.equal(e, want)
    n_new = 4
    call = dict(input_ids=ids, attention_mask=torch.ones_like(ids), do_sample=True, max_new_tokens=n_new, min_new_tokens=1, length_penalty=1,
                num_return_sequences=1, output_hidden_states=True, use_cache=True, pad_token_id=eod, eos_token_id=eod, temperature=1.0, top_p=None,
                top_k=None, images=img, cd_beta=0.1, cd_alpha=1.0, output_scores=True, return_dict_in_generate=True, cd_greedy=True)

    class RefQwenLM(RefLavisLM):
        def prepare_inputs_for_generation_cd(self, input_ids, **kw):        # modeling_qwen.py:1089-1118: same inputs, images = images_cd
            d = self.prepare_inputs_for_generation(input_ids, **kw)
            if kw.get("images_cd") is not None and "inputs_embeds" in d:
                d["inputs_embeds"] = kw["images_cd"]
            return d
    ref = RefQwenLM(eng.w, device=DEV)
    e_cd = qwen_spliced_embeddings(model, ids, img_cd)[0]
    for mode, kw_ref in (("dd_unk", dict(use_dd_unk=True)), ("vcd", dict(images_cd=e_cd[None].float().cpu()))):
        got = model.generate(**call, images_cd=img_cd if mode == "vcd" else None, use_dd=False, use_dd_unk=mode == "dd_unk")
        L = ids.shape[1]
        assert got["sequences"].shape == (1, L + n_new) and torch.equal(got["sequences"][:, :L], ids)
        assert torch.isneginf(got["scores"][0][0, eod])                  # min_new_tokens = 1
        r = O.reference_loop(ref, torch.zeros(1, 0, dtype=torch.long), warp=O.WarpConfig(temperature=1.0), max_length=n_new, pad_token_id=eod,
                             eos_token_id=eod, pick=O.pick_argmax, processors=O.ProcessorList([O.MinNewTokens(0, 1, [eod])]),
                             inputs_embeds=e[None].float().cpu(), attention_mask=torch.ones(1, e.shape[0], dtype=torch.long), use_cache=True,
                             cd_alpha=1.0, cd_beta=0.1, **kw_ref)
        _check_rows(got["scores"], got["tokens"], [r], n_new)            # scores within the dtype's noise, tokens where the margin clears it
        assert (got["tokens"][0].cpu() == r.sequences[0]).float().mean().item() >= 0.75, mode
    with pytest.raises(ValueError, match="rotary table"):
        model.generate(**dict(call, max_new_tokens=600))                  # beyond seq_length the reference switches to NTK scaling: refused
    detach_engine(model)
