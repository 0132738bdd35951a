"""VddLlavaEngine.generate_list: a question LIST with a bounded number of questions in flight - waiting questions are prefilled into the
slots of finished ones while the rest keeps decoding (the reference walks its list one B = 1 generate() call at a time,
llava_sampling.py:78-116 / llava_calibrate.py:130, so a question's answer never depends on the others: the checker below)."""
import numpy as np
import pytest
import torch

from test_engine_shapes_gpu import _engine, _prompts

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
W7B = dict(d=4096, n_heads=32, n_kv_heads=32, head_dim=128, ffn=11008, vocab=32000)


def _eos_set(n, seed, vocab=32000):
    return sorted(set(np.random.default_rng(seed).integers(3, vocab, size=n).tolist()))


def _answer_lengths(tokens, eos):
    eos_t = torch.tensor(eos, device=tokens.device)
    is_eos = (tokens[:, :, None] == eos_t).any(-1)
    return torch.where(is_eos.any(1), is_eos.float().argmax(1) + 1, torch.full((tokens.shape[0],), tokens.shape[1], device=tokens.device))


@pytest.mark.parametrize("mode", [dict(use_dd_unk=True, temperature=0.5), dict(use_dd=True, use_dd_unk=True, temperature=1.0, top_p=0.9)], ids=["dd_unk", "both_top_p"])
def test_list_answers_equal_the_batch_answers_in_batch_invariant_mode(mode):
    """40 questions, 12 in flight, answers of 1 ... 48 tokens (a random EOS set): every answer, its padding and its step-0 top-10 are those
    of ONE generate() call over all 40 questions - token for token under cd_greedy in batch-invariant mode - although every question
    decoded next to different neighbours, in a reused slot, from a step index of its own."""
    from llava_align_amd import ops
    eng = _engine(W7B, n_layers=3, vit_layers=2)
    ids, imgs = _prompts(40, 1, 32000, seed=91)
    eos = _eos_set(900, 5)                                   # ~3 % of the vocabulary: geometric answer lengths, mean ~ 30 tokens, capped at 48
    kw = dict(cd_alpha=1.0, cd_beta=0.1, cd_greedy=True, max_new_tokens=48, eos_token_id=eos, pad_token_id=0, n_top=10, **mode)
    with ops.batch_invariant():
        eng.retire = False
        ref = eng.generate(ids, images=imgs, sync_every=4, **kw)
        out = eng.generate_list(ids, imgs, in_flight=12, sync_every=4, admit_min=2, **kw)
    la, lb = _answer_lengths(ref.tokens, eos), _answer_lengths(out.tokens, eos)
    assert torch.equal(la, lb) and int(la.min()) < 10 and int(la.max()) > 30                    # the answers really have different lengths
    T = min(ref.tokens.shape[1], out.tokens.shape[1])
    assert torch.equal(ref.tokens[:, :T], out.tokens[:, :T])
    for q in range(40):                                                                        # padded like the reference pads (vcd_sample.py:260)
        assert bool((out.tokens[q, int(lb[q]):] == 0).all())
    assert torch.equal(ref.top_tok, out.top_tok) and torch.equal(ref.top_prob, out.top_prob)
    st = out.stats
    assert st["admissions"] >= 4 and st["in_flight"] == 12 and st["questions"] == 40 and st["graph"] and st["answer_tokens"] == int(lb.sum())
    assert st.get("tail_shrinks", 0) >= 1                                                      # the end of the list ran on a smaller step
    assert [s.shape[0] for s in out.sequences] == [len(i) + out.tokens.shape[1] for i in ids]


@pytest.mark.parametrize("in_flight", [8, 3])                # 16 rows / 6 rows: the few-row layer forms decode the list too
def test_sampled_list_run_is_reproducible_and_well_formed(in_flight):
    eng = _engine(W7B, n_layers=2, vit_layers=2)
    ids, imgs = _prompts(5, 6, 32000, seed=17)               # 30 questions, 6 per image: image prefixes shared across slots and waves
    eos = _eos_set(1500, 3)
    kw = dict(in_flight=in_flight, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.7, max_new_tokens=24, eos_token_id=eos, pad_token_id=0, seed=11,
              sync_every=2)
    a = eng.generate_list(ids, imgs, **kw)
    b = eng.generate_list(ids, imgs, **kw)
    assert torch.equal(a.tokens, b.tokens)                                                      # same seed, same schedule of admissions
    c = eng.generate_list(ids, imgs, **dict(kw, seed=12))
    assert not torch.equal(a.tokens[:, : min(a.tokens.shape[1], c.tokens.shape[1])], c.tokens[:, : min(a.tokens.shape[1], c.tokens.shape[1])])
    L = _answer_lengths(a.tokens, eos)
    eos_t = torch.tensor(eos, device=DEV)
    for q in range(30):
        n = int(L[q])
        assert n == a.tokens.shape[1] or n == 24 or bool((a.tokens[q, n - 1] == eos_t).any())   # ends with EOS unless it ran to the cap
        assert bool((a.tokens[q, n:] == 0).all())
    assert a.stats["admissions"] >= 3 and 0 < a.stats["mean_live_rows"] <= 2 * in_flight


def test_list_refuses_what_it_does_not_do():
    eng = _engine(W7B, n_layers=1, vit_layers=1)
    ids, imgs = _prompts(2, 1, 32000, seed=1)
    with pytest.raises(ValueError, match="eos_token_id"):
        eng.generate_list(ids, imgs, max_new_tokens=4)
    with pytest.raises(ValueError, match="pad_token_id"):
        eng.generate_list(ids, imgs, max_new_tokens=4, eos_token_id=2)
    with pytest.raises(ValueError, match="one image placeholder"):
        eng.generate_list([torch.tensor([1, 5, 6])], imgs[:1], max_new_tokens=4, eos_token_id=2, pad_token_id=0)


def test_retirement_carries_the_in_kernel_processors():
    """Round 6: a call with min_new_tokens / repetition_penalty (the Qwen MME and InstructBLIP call shapes, run_qwen.py:194,
    blip2_vicuna_instruct.py:400) retires finished rows too - the EOS floor and the penalty history are per-question state that moves with
    its question.  Token for token equal to the static run in batch-invariant mode."""
    from llava_align_amd import ops
    eng = _engine(W7B, n_layers=2, vit_layers=2)
    ids, _ = _prompts(24, 1, 32000, seed=41)
    ids = [torch.tensor([t for t in r.tolist() if t != -200]) for r in ids]       # slot-free prompts: what the reference combines the penalty with (LAVIS, Qwen)
    eos = _eos_set(600, 7)
    kw = dict(temperature=1.0, top_p=0.9, cd_greedy=True, max_new_tokens=160, eos_token_id=eos, pad_token_id=0, min_new_tokens=6, repetition_penalty=1.2,
              sync_every=8)
    with ops.batch_invariant():
        eng.retire, eng.kv_chunk = False, 32
        a = eng.generate(ids, **kw)
        eng._kvs.clear(); eng._graphs.clear()
        eng.retire = True
        b = eng.generate(ids, **kw)
    assert "retire_events" not in a.stats and b.stats["retire_events"] >= 2 and b.stats["rows_at_end"] < 24
    assert int(_answer_lengths(a.tokens, eos).min()) >= 6                                       # the EOS floor held
    assert a.tokens.shape == b.tokens.shape and torch.equal(a.tokens, b.tokens)


def test_run_sampling_is_shard_and_in_flight_invariant_for_deterministic_decodes():
    """sampling_driver.run_sampling (the batched form of llava_sampling.py:57-126) with cd_greedy: the driver selects batch-invariant mode, so the
    answers do not depend on how many questions are in flight - and equal ONE generate() call over the list."""
    from llava_align_amd import ops
    from llava_align_amd.sampling_driver import run_sampling
    eng = _engine(W7B, n_layers=2, vit_layers=2)
    ids, imgs = _prompts(14, 1, 32000, seed=33)
    images = {f"im{i}.jpg": imgs[i] for i in range(14)}
    by_text = {f"q{i}": ids[i].tolist() for i in range(14)}
    qs = [{"question_id": 50 + i, "image": f"im{i}.jpg", "text": f"q{i}"} for i in range(14)]
    enc = lambda prompt: by_text[prompt[prompt.index("<image>\n") + 8: prompt.rindex(" ASSISTANT:")]]
    dec = lambda t: " ".join(map(str, t))
    eos = _eos_set(1200, 9)
    kw = dict(max_new_tokens=20, eos_token_id=eos, pad_token_id=0, stop_str=None, use_dd=True, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=1.0,
              top_p=0.9, cd_greedy=True)
    a = run_sampling(eng, qs, enc, dec, lambda n: images[n], in_flight=4, **kw)
    b = run_sampling(eng, qs, enc, dec, lambda n: images[n], in_flight=14, **kw)
    assert a["batch_invariant"] and [x["text"] for x in a["answers"]] == [x["text"] for x in b["answers"]]
    assert a["stats"]["admissions"] >= 3 and b["stats"]["admissions"] == 1
    with ops.batch_invariant():
        eng.retire = False
        ref = eng.generate(ids, images=imgs, **{k: v for k, v in kw.items() if k != "stop_str"})
    eos_s = set(eos)
    for i, x in enumerate(a["answers"]):
        want = []
        for t in ref.tokens[i].tolist():
            want.append(t)
            if t in eos_s:
                break
        assert x["text"] == " ".join(map(str, want)), i


def test_in_flight_is_lowered_to_what_the_device_holds(monkeypatch):
    """The own-KV slots of a list run are held at full length: with little memory free the call lowers the number in flight (by eighths, not
    below 8) instead of failing in the allocator, says so in its stats, and still answers every question (here: equal to the batch run)."""
    from llava_align_amd import ops
    from llava_align_amd.engine import KVCache
    eng = _engine(W7B, n_layers=2, vit_layers=2)
    ids, imgs = _prompts(36, 1, 32000, seed=23)
    eos = _eos_set(1500, 7)
    kw = dict(use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, cd_greedy=True, temperature=0.5, max_new_tokens=32, eos_token_id=eos, pad_token_id=0)
    with ops.batch_invariant():
        ref = eng.generate(ids, images=imgs, **kw)
        eng._kvs.clear(); eng._graphs.clear(); eng._kv = None
        lm = eng.cfg.lm
        # room for the pools of ~16 questions in flight: 32 asked -> 28 -> 24 -> 21 -> 18 -> 15
        room = KVCache.bytes_needed(lm, 16 + 2, 640, 2 * 16, 64 + 32, eng.dtype, False) / eng.KV_MEMORY_FRACTION
        real = torch.cuda.mem_get_info
        monkeypatch.setattr(torch.cuda, "mem_get_info", lambda d=None: (int(room) - (torch.cuda.memory_reserved(d) - torch.cuda.memory_allocated(d)), real(d)[1]))
        out = eng.generate_list(ids, imgs, in_flight=32, **kw)
    assert out.stats["in_flight_asked"] == 32 and 8 <= out.stats["in_flight"] <= 16 and out.stats["n_rows"] == 2 * out.stats["in_flight"]
    T = min(ref.tokens.shape[1], out.tokens.shape[1])
    assert torch.equal(ref.tokens[:, :T], out.tokens[:, :T])


@pytest.mark.parametrize("both", [False, True], ids=["vcd", "vcd_plus_both_dd"])
def test_vcd_list_answers_equal_the_batch_answers(both):
    """images_cd through generate_list: the cd prompt of every admitted question (tokens | noised patches | suffix) is prefilled as one
    sequence into a scratch prefix slot, contrasts step 0, and never decodes (from step 1 on the reference's cd pass reads the main cache:
    c == v, quirk #1).  Token for token and top-10 for top-10 the answers of ONE generate(images_cd=...) call in batch-invariant mode."""
    from llava_align_amd import ops
    eng = _engine(W7B, n_layers=3, vit_layers=2)
    ids, imgs = _prompts(26, 1, 32000, seed=47)
    g = torch.Generator().manual_seed(9)
    imgs_cd = [im + 0.8 * torch.randn(im.shape, generator=g).to(im.device, im.dtype) for im in imgs]
    eos = _eos_set(900, 5)
    kw = dict(cd_alpha=1.0, cd_beta=0.1, cd_greedy=True, temperature=0.7, max_new_tokens=40, eos_token_id=eos, pad_token_id=0, n_top=10,
              use_dd=both, use_dd_unk=both)
    with ops.batch_invariant():
        eng.retire = False
        ref = eng.generate(ids, images=imgs, images_cd=imgs_cd, sync_every=4, **kw)
        plain = eng.generate(ids, images=imgs, sync_every=4, **kw)
        out = eng.generate_list(ids, imgs, images_cd=imgs_cd, in_flight=8, sync_every=4, admit_min=2, **kw)
    la, lb = _answer_lengths(ref.tokens, eos), _answer_lengths(out.tokens, eos)
    assert torch.equal(la, lb) and int(la.max()) > 20
    T = min(ref.tokens.shape[1], out.tokens.shape[1])
    assert torch.equal(ref.tokens[:, :T], out.tokens[:, :T])
    assert torch.equal(ref.top_tok, out.top_tok) and torch.equal(ref.top_prob, out.top_prob)
    assert not torch.equal(ref.top_prob, plain.top_prob)                                       # the noised branch really contrasts step 0
    st = out.stats
    assert st["n_rows"] == (2 if both else 1) * 8 and st["admissions"] >= 3 and st["graph"]


@pytest.mark.parametrize("mode", ["plain", "dd_unk", "vcd"])
def test_embedding_prompts_through_the_list(mode):
    """The Qwen-VL / InstructBLIP call shape: prompts as [T, d] embeddings, `embeds_prefix` = rows two questions about one image share, the
    image-free branch re-running the same inputs, `images_cd` = the noisy-image embeddings of the whole prompt.  Answers and
    step-0 top-10 equal ONE generate(inputs_embeds=...) call in batch-invariant mode."""
    from llava_align_amd import ops
    eng = _engine(W7B, n_layers=3, vit_layers=2)
    g = torch.Generator(device=DEV).manual_seed(3)
    mk = lambda n: (torch.randn(n, 4096, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    shared = [mk(40) for _ in range(9)]                                       # '<img>' + image slots: one per image, two questions each
    emb = [torch.cat([shared[i // 2], mk(7 + i % 5)]) for i in range(18)]
    pre = [(f"im{i // 2}", 40) for i in range(18)]
    emb_cd = [torch.cat([mk(40), e[40:]]) for e in emb]
    eos = _eos_set(900, 5)
    kw = dict(cd_alpha=1.0, cd_beta=0.1, cd_greedy=True, temperature=0.7, max_new_tokens=36, eos_token_id=eos, pad_token_id=0, n_top=10, use_dd_unk=mode == "dd_unk")
    extra = dict(images_cd=emb_cd) if mode == "vcd" else {}
    with ops.batch_invariant():
        eng.retire = False
        ref = eng.generate(None, inputs_embeds=emb, embeds_prefix=pre, sync_every=4, **kw, **extra)
        out = eng.generate_list(None, inputs_embeds=emb, embeds_prefix=pre, in_flight=6, sync_every=4, admit_min=2, **kw, **extra)
        bare = eng.generate_list(None, inputs_embeds=emb, in_flight=6, sync_every=4, admit_min=2, **kw, **extra)     # no declared sharing: same answers
    la, lb = _answer_lengths(ref.tokens, eos), _answer_lengths(out.tokens, eos)
    assert torch.equal(la, lb) and int(la.max()) > 15
    T = min(ref.tokens.shape[1], out.tokens.shape[1], bare.tokens.shape[1])
    assert torch.equal(ref.tokens[:, :T], out.tokens[:, :T]) and torch.equal(ref.tokens[:, :T], bare.tokens[:, :T])
    assert torch.equal(ref.top_tok, out.top_tok) and torch.equal(ref.top_prob, out.top_prob)
    assert out.stats["n_rows"] == (2 if mode == "dd_unk" else 1) * 6 and out.stats["admissions"] >= 3
    assert out.stats["prefill_tokens"] < bare.stats["prefill_tokens"] or mode == "dd_unk"          # the declared image rows were prefilled once per image
    assert [s_.shape[0] for s_ in out.sequences] == [out.tokens.shape[1]] * 18
    with pytest.raises(ValueError, match="replace input_ids"):
        eng.generate_list([torch.tensor([1, -200, 5])], None, inputs_embeds=emb[:1], eos_token_id=eos, pad_token_id=0)


def test_short_lists_shared_images_and_one_question():
    """Edges: fewer questions than slots, ONE question, six questions per image with the VCD branch (the clean prefix is shared across slots
    and waves, every cd prompt has its own noise), a list shorter than one admission wave - each against generate() in batch-invariant mode."""
    from llava_align_amd import ops
    eng = _engine(W7B, n_layers=2, vit_layers=2)
    ids, imgs = _prompts(3, 6, 32000, seed=29)                               # 18 questions, 6 per image
    g = torch.Generator().manual_seed(4)
    imgs_cd = [im + 0.5 * torch.randn(im.shape, generator=g).to(im.device, im.dtype) for im in imgs]
    eos = _eos_set(1200, 8)
    kw = dict(cd_alpha=1.0, cd_beta=0.1, cd_greedy=True, temperature=0.6, max_new_tokens=20, eos_token_id=eos, pad_token_id=0)
    with ops.batch_invariant():
        eng.retire = False
        for sel, extra, fl in ((range(18), dict(images_cd=imgs_cd), 5), (range(3), dict(use_dd_unk=True), 8), ([7], dict(use_dd=True, use_dd_unk=True), 4),
                               ([4], dict(images_cd=imgs_cd), 1)):
            sel = list(sel)
            ex = {k: ([v[i] for i in sel] if k == "images_cd" else v) for k, v in extra.items()}
            ref = eng.generate([ids[i] for i in sel], images=[imgs[i] for i in sel], **kw, **ex)
            out = eng.generate_list([ids[i] for i in sel], [imgs[i] for i in sel], in_flight=fl, sync_every=2, **kw, **ex)
            T = min(ref.tokens.shape[1], out.tokens.shape[1])
            assert torch.equal(ref.tokens[:, :T], out.tokens[:, :T]), (sel, list(extra))
            assert out.stats["in_flight"] == min(fl, len(sel)) and out.stats["questions"] == len(sel)
            for q, i in enumerate(sel):
                assert torch.equal(out.sequences[q][: len(ids[i])].cpu(), ids[i]) and out.sequences[q].shape[0] == len(ids[i]) + out.tokens.shape[1]
    with pytest.raises(ValueError, match="eos_token_id"):
        eng.generate_list(ids[:2], imgs[:2], max_new_tokens=4)
    with pytest.raises(ValueError, match="one images_cd entry per question"):
        eng.generate_list(ids[:2], imgs[:2], images_cd=imgs_cd[:1], eos_token_id=eos, pad_token_id=0)
