"""Config #4 / #5 drivers over the engine: mme_driver.run_mme (LLaVA and Qwen call shapes, run_llava.py / run_qwen.py +
convert_answer_to_mme_calibrate.py + calculation.py) and blip_driver.run_blip_pope (blip_calibrate.py), against the reference's
per-question procedure restated with the oracle loop over the fp32 reference model."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import vdd_oracle as O
from ref_llava import RefLlava

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CATS = ["existence", "count", "position", "color", "commonsense_reasoning", "numerical_calculation", "text_translation", "code_reasoning"]


def decode_token(t):
    return {0: "yes", 1: " Yes", 2: "no", 3: "No "}.get(t % 11, f"w{t}")


def decode(ids):
    return " ".join(decode_token(t).strip() for t in ids)


def toy_encode(prompt):
    """'<image>' -> -200; every other word -> a stable id in [3, 1000)."""
    out = []
    for w in prompt.replace("<image>", " <image> ").split():
        out.append(-200 if w == "<image>" else (sum(ord(c) * (i + 1) for i, c in enumerate(w)) % 997) + 3)
    return [1] + out


def mme_questions(n_img=2):
    qs, gt = [], {}
    for ci, cat in enumerate(CATS):
        for im in range(n_img):
            for k in range(2):
                text = f"Is thing {ci}{im}{k} here?"
                qs.append({"question_id": f"{cat}/{im:04d}.png", "image": f"{cat}/{im:04d}.png", "category": cat,
                           "text": text + "\nAnswer the question using a single word or phrase."})
                gt[(cat, f"{im:04d}.txt", text + " Please answer yes or no.")] = ("Yes", "No")[(ci + im + k) % 2]
    qs.append({"question_id": "artwork/0001.png", "image": "artwork/0001.png", "category": "artwork", "text": "filtered out"})
    return qs, gt


@pytest.fixture(scope="module")
def eng():
    from llava_align_amd.engine import LlavaWeights, VddLlavaEngine, preset
    cfg = preset("tiny")
    return VddLlavaEngine(cfg, weights=LlavaWeights.random(cfg, DEV, seed=3, std=0.06), device=DEV, use_graph=False)


def test_run_mme_llava_call_shape_against_the_per_question_procedure(eng, tmp_path):
    from llava_align_amd import calibrate as C
    from llava_align_amd.mme_driver import ONE_WORD, llava_mme_inputs, run_mme, vicuna_v1_prompt
    ref = RefLlava(eng.w, device=DEV)
    qs, gt = mme_questions()
    images = {}

    def load_image(name):
        if name not in images:
            images[name] = torch.randn(3, 56, 56, generator=torch.Generator().manual_seed(len(images)))
        return images[name]
    build = llava_mme_inputs(toy_encode, load_image, unk_token_id=0)
    res = run_mme(eng, qs, build, decode, answers_path=str(tmp_path / "a" / "ans.jsonl"), batch_questions=12, max_new_tokens=3,
                  gt=gt, results_root=str(tmp_path / "res"), experiment="tiny", use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1,
                  temperature=0.5, cd_greedy=True)
    lines = [json.loads(l) for l in open(tmp_path / "a" / "ans.jsonl")]
    assert len(lines) == 32 and [l["question_id"] for l in lines] == [q["question_id"] for q in qs[:-1]]       # 'artwork' is filtered
    assert set(lines[0]) == {"question_id", "prompt", "text", "naive", "none", "unk", "answer_id", "model_id", "metadata"}
    assert set(res["results"]) == {"naive", "none", "unk", "none_unk"}
    for name, d in res["results"].items():
        assert sorted(os.listdir(d)) == sorted(c + ".txt" for c in CATS)
        assert res["scores"][name] is not None and 0 <= res["scores"][name]["Perception"]["total"] <= 800
    assert res["converted"] == C.mme_convert(lines, gt)

    def step0(ids, img, **kw):
        kw = dict(images=img[None] if img is not None else None, attention_mask=torch.ones(1, len(ids), dtype=torch.long), use_cache=True,
                  cd_alpha=1.0, cd_beta=0.1, **kw)
        r = O.reference_loop(ref, torch.tensor([ids]), warp=O.WarpConfig(temperature=0.5), max_length=len(ids) + 1, pad_token_id=None,
                             eos_token_id=None, pick=O.pick_argmax, **kw)
        tp, tt = torch.topk(torch.softmax(r.scores[0][0].float(), -1), 10)
        return C.label_dict_from_top(tt.tolist(), tp.tolist(), decode_token)
    for q, a in list(zip(qs, lines))[::5]:
        text = q["text"]
        ids_main = toy_encode(vicuna_v1_prompt("<image>\n" + text))
        ids_unk = [0 if t == -200 else t for t in toy_encode(vicuna_v1_prompt("<image>\n" + text + ONE_WORD))]
        want = (step0(ids_main, images[q["image"]], use_dd_unk=True), step0(toy_encode(vicuna_v1_prompt(text + ONE_WORD)), None),
                step0(ids_unk, None))
        for got, w in zip((a["naive"], a["none"], a["unk"]), want):
            pg, pw = np.array(C.get_prob_from_logits(got)), np.array(C.get_prob_from_logits(w))
            assert np.abs(pg - pw).max() <= 0.05 + 0.15 * pw.max(), (q["question_id"], pg, pw)


def test_run_mme_qwen_call_shape_dual_pass_and_calibrate():
    """run_qwen.py:190-221: prompts as embeddings (256 image slots + text), min_new_tokens=1, pad = eos = eod id, use_dd_unk whose
    image-free branch re-runs the SAME inputs, then the two text-only prior passes and the affine calibration."""
    from llava_align_amd import calibrate as C
    from llava_align_amd.engine import LlavaWeights, VddLlavaEngine, preset
    from llava_align_amd.mme_driver import qwen_mme_inputs, run_mme
    cfg = preset("tiny-qwen")
    eng = VddLlavaEngine(cfg, weights=LlavaWeights.random(cfg, DEV, seed=5, std=0.06), device=DEV, use_graph=True)
    qs, gt = mme_questions(n_img=1)
    table = eng.w.t["embed"]
    g = torch.Generator(device=DEV).manual_seed(1)
    feats = {}

    def embed_prompt(text, path):
        ids = torch.tensor([t for t in toy_encode(text) if t >= 0], device=DEV)
        e = table[ids]
        if path is not None:                                          # 16 "resampler" slots stand in for Qwen's 256
            if path not in feats:
                feats[path] = (torch.randn(16, cfg.lm.d, device=DEV, generator=g) * 0.06).to(torch.bfloat16)
            e = torch.cat([e[:1], feats[path], e[1:]], 0)
        return e
    eod = 151643
    res = run_mme(eng, qs, qwen_mme_inputs(embed_prompt), decode, batch_questions=8, max_new_tokens=4, min_new_tokens=1, eos_token_id=eod,
                  pad_token_id=eod, gt=gt, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, cd_greedy=True, output_scores=True)
    assert len(res["answers"]) == 16 and set(res["converted"]) == {"naive", "none", "unk", "none_unk"}
    # the dual pass really ran two rows per question with identical inputs; their contrast keeps exactly the beta-plausible set of v
    emb = [embed_prompt("<img>{}</img>{} Answer:".format(q["image"], q["text"]), q["image"]) for q in qs[:4]]
    dd = eng.generate(None, inputs_embeds=emb, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, cd_greedy=True, max_new_tokens=2,
                      output_scores=True, min_new_tokens=1, eos_token_id=eod, pad_token_id=eod)
    plain = eng.generate(None, inputs_embeds=emb, temperature=0.5, cd_greedy=True, max_new_tokens=2, output_scores=True)
    assert dd.stats["n_rows"] == 8 and plain.stats["n_rows"] == 4
    # both branches are fed by the same embeddings: their prompt K/V are computed once (everything but the last position is a shared prefix)
    assert dd.stats["prefill_tokens"] == plain.stats["prefill_tokens"] + 4 and dd.stats["unshared_prefill_tokens"] == 2 * plain.stats["prefill_tokens"]
    s_dd, s_pl = dd.scores[0].float(), plain.scores[0].float()
    assert torch.isneginf(s_dd[:, eod]).all()                         # min_new_tokens=1 at step 0
    keep = torch.isfinite(s_dd)
    cut = s_pl.max(-1, keepdim=True).values + np.log(0.1) / 0.5
    assert ((s_pl >= cut + 0.3) & ~keep).sum() <= 4 and ((s_pl < cut - 0.3) & keep).sum() == 0     # beta mask of v (eos aside)
    assert (s_dd[keep] - s_pl[keep]).abs().max() <= 0.3               # (1+a) v - a c with c ~ v  ->  v


def test_run_qwen_pope_five_passes_against_direct_calls(tmp_path):
    """qwen_calibrate.py:90-168 on a QWenLMHeadModel-shaped object (its `transformer.visual` fills the <img> span): main pass with
    use_dd_unk + four content-free priors; the answers file's fields; the deterministic priors (none / unk / zero) recomputed one question
    at a time; image spans shared between the questions about one image; the answers-only call shape of qwenvl_sampling.py."""
    import hf_doubles
    from llava_align_amd import calibrate as C
    from llava_align_amd.checkpoint import qwen_embed_prompt
    from llava_align_amd.engine import VddLlavaEngine
    from llava_align_amd.hf_adapter import lm_config_from_hf, lm_weights_from_hf
    from llava_align_amd.qwen_driver import CALIBRATE_NAMES, SAMPLING_PROMPT, run_qwen_pope
    model = hf_doubles.build_qwen(DEV, torch.bfloat16)
    cfg = lm_config_from_hf(model)
    eng = VddLlavaEngine(cfg, weights=lm_weights_from_hf(model, cfg), device=DEV)
    V, eod, st = model.config.vocab_size, model.generation_config.eos_token_id, model.config.visual["image_start_id"]
    visual_calls = []
    model.transformer.visual.register_forward_hook(lambda m, a, o: visual_calls.append(a[0].shape[0]))

    def tokenize(text):                                   # '<img>path</img>' -> <img> + img_rows slots + </img>, as the Qwen tokenizer expands it
        out = []
        for i, part in enumerate(text.replace("</img>", "<img>").split("<img>")):
            out += [st] + [st + 2] * model.img_rows + [st + 1] if i % 2 else hf_doubles.word_ids(part, V - 40)
        return out
    embed = qwen_embed_prompt(model, tokenize, DEV)
    images = {f"im{i}.jpg": torch.randn(3, 16, 16, generator=torch.Generator().manual_seed(40 + i)) for i in range(3)}
    qs = [{"question_id": i, "image": f"im{i // 3}.jpg", "text": f"Is there a thing{i} in the image?", "label": ("yes", "no")[i % 2]} for i in range(9)]
    torch.manual_seed(3)
    kw = dict(eos_token_id=eod, pad_token_id=eod, max_new_tokens=4, temperature=0.5, cd_alpha=1.0, cd_beta=0.1)
    res = run_qwen_pope(eng, qs, embed, decode, lambda n: images[n], image_path=lambda f: "/imgs/" + f, answers_path=str(tmp_path / "q" / "qwen.jsonl"),
                        batch_questions=6, use_dd_unk=True, cd_greedy=True, **kw)
    lines = [json.loads(l) for l in open(tmp_path / "q" / "qwen.jsonl")]
    assert [l["question_id"] for l in lines] == list(range(9))
    assert list(lines[0]) == ["question_id", "prompt", "text", "naive", "noise", "none", "zero", "unk", "model_id", "image", "metadata"]     # :155-166
    assert lines[4]["prompt"] == "<img>/imgs/im1.jpg</img>Is there a thing4 in the image? Answer:" and lines[0]["model_id"] == "qwen-vl"
    assert set(res["scores"]) == {"string_match"} | set(CALIBRATE_NAMES) and res["batch_invariant"]
    # the tower ran once per clean image, once per question for the fresh noise of the `noise` prior, once per batch for the zero image
    assert sorted(visual_calls) == [1] * (3 + 9 + 2)

    def step0(text, image):
        e = embed(text, image)
        o = eng.generate(None, inputs_embeds=[e[0] if isinstance(e, tuple) else e], max_new_tokens=1, min_new_tokens=1, n_top=10, eos_token_id=eod,
                         pad_token_id=eod, temperature=0.5)
        return C.label_dict_from_top(o.top_tok[0].tolist(), o.top_prob[0].tolist(), decode_token)
    zero = torch.zeros(3, 16, 16)
    for q, a in zip(qs, lines):
        want = {"none": step0("{} Answer:".format(q["text"]), None), "unk": step0("None {} Answer:".format(q["text"]), None),
                "zero": step0(a["prompt"], zero)}
        for name, w in want.items():
            pg, pw = np.array(C.get_prob_from_logits(a[name])), np.array(C.get_prob_from_logits(w))
            assert np.abs(pg - pw).max() <= 0.02 + 0.05 * pw.max(), (q["question_id"], name, pg, pw)
        assert a["noise"] != a["zero"] and len(a["naive"]) >= 1
    # main pass == the engine called directly on the same embeddings (cd_greedy: deterministic), incl. the EOS floor of min_new_tokens = 1
    emb = [embed(l["prompt"], images[q["image"]])[0] for q, l in zip(qs, lines)]
    from llava_align_amd import ops
    with ops.batch_invariant():
        direct = eng.generate(None, inputs_embeds=emb, use_dd_unk=True, cd_greedy=True, min_new_tokens=1, **kw)
    from llava_align_amd.pope_driver import cut_at_eos
    assert [l["text"] for l in lines] == [decode(cut_at_eos(t, {eod})).strip() for t in direct.tokens.tolist()]
    # VCD: images_cd = the tower's rows for a noised copy, per question; answers-only shape (no priors, no label dicts in the file)
    n0 = len(visual_calls)
    s = run_qwen_pope(eng, qs[:3], embed, decode, lambda n: images[n], answers_path=str(tmp_path / "q" / "s.jsonl"), priors=(), prompt_format=SAMPLING_PROMPT,
                      use_cd=True, noise_step=500, seed=5, **kw)
    assert len(visual_calls) - n0 == 1 + 3 and s["scores"].keys() == {"string_match"} and not s["batch_invariant"]
    first = json.loads(open(tmp_path / "q" / "s.jsonl").readline())
    assert list(first) == ["question_id", "prompt", "text", "model_id", "image", "metadata"] and first["prompt"].startswith("Question: <img>im0.jpg</img> Is")
    # a sampling sweep over the POPE file (qwenvl_sampling.py:147-185): three settings decode every batch from ONE prefill; each equals its own run
    sw = [dict(tag="default", temperature=1.0, top_p=None, top_k=None, answers_path=str(tmp_path / "q" / "sw-default.jsonl")),
          dict(tag="temp_0.3", temperature=0.3, top_p=None, top_k=None), dict(tag="top_k_2", temperature=1.0, top_p=None, top_k=2)]
    skw = {k: v for k, v in kw.items() if k != "temperature"}
    reused = []
    real = eng.generate
    eng.generate = lambda *a_, **k_: (lambda o_: reused.append(bool(o_.stats.get("prefill_reused"))) or o_)(real(*a_, **k_))
    try:
        swept = run_qwen_pope(eng, qs, embed, decode, lambda n: images[n], priors=(), prompt_format=SAMPLING_PROMPT, batch_questions=6, use_dd_unk=True,
                              cd_greedy=True, sweep=sw, **skw)
    finally:
        eng.generate = real
    assert reused == [False, True, True] * 2                   # two batches: one prefill, two decodes from it
    for s_ in sw:
        alone = run_qwen_pope(eng, qs, embed, decode, lambda n: images[n], priors=(), prompt_format=SAMPLING_PROMPT, batch_questions=6, use_dd_unk=True,
                              cd_greedy=True, temperature=s_["temperature"], top_k=s_["top_k"], **skw)
        assert [a_["text"] for a_ in swept["runs"][s_["tag"]]["answers"]] == [a_["text"] for a_ in alone["answers"]], s_["tag"]
    assert len(open(sw[0]["answers_path"]).readlines()) == 9
    with pytest.raises(ValueError, match="answers-only"):
        run_qwen_pope(eng, qs, embed, decode, lambda n: images[n], sweep=sw, **skw)
    # open-ended shape (no EOS floor): the shard goes through generate_list, 4 in flight - same answers as one generate() call over all nine
    eos = sorted(set(np.random.default_rng(2).integers(3, V - 40, size=60).tolist()))
    lkw = dict(kw, eos_token_id=eos, pad_token_id=eod, max_new_tokens=24, min_new_tokens=None, cd_greedy=True, use_dd_unk=True)
    lst = run_qwen_pope(eng, qs, embed, decode, lambda n: images[n], priors=(), prompt_format=SAMPLING_PROMPT, batch_questions=4, **lkw)
    assert lst["stats"]["in_flight"] == 4 and lst["stats"]["admissions"] >= 2 and lst["batch_invariant"]
    emb = [embed(a["prompt"], images[q["image"]])[0] for q, a in zip(qs, lst["answers"])]
    with ops.batch_invariant():
        direct = eng.generate(None, inputs_embeds=emb, **{k: v for k, v in lkw.items() if k != "min_new_tokens"})
    assert [a["text"] for a in lst["answers"]] == [decode(cut_at_eos(t, set(eos))).strip() for t in direct.tokens.tolist()]


def test_run_blip_pope_vcd_against_direct_front_end_and_engine_calls(tmp_path):
    from llava_align_amd import calibrate as C
    from llava_align_amd.blip_driver import QUESTION_SUFFIX, map_pad_to_eos, run_blip_pope
    from llava_align_amd.blip_frontend import BlipWeights, InstructBlipFrontEnd, tiny_blip_config
    from llava_align_amd.engine import LlavaWeights, VddLlavaEngine, preset
    cfg = preset("tiny")
    eng = VddLlavaEngine(cfg, weights=LlavaWeights.random(cfg, DEV, seed=3, std=0.06), device=DEV, use_graph=False)
    front = InstructBlipFrontEnd(BlipWeights.random(tiny_blip_config(), DEV, seed=4, std=0.05))
    images = {f"im{i}.jpg": torch.randn(3, 56, 56, generator=torch.Generator().manual_seed(70 + i)) for i in range(2)}
    qs = [{"question_id": i, "image": f"im{i % 2}.jpg", "text": f"Is there a thing{i}?", "label": ("yes", "no")[i % 2]} for i in range(5)]
    tok_llm = lambda p: [t % 997 + 3 for t in toy_encode(p)]
    tok_qf = lambda p: [101] + [t % 400 + 5 for t in toy_encode(p)[1:9]] + [102]
    torch.manual_seed(11)
    res = run_blip_pope(eng, front, qs, tok_llm, tok_qf, decode, lambda n: images[n], answers_path=str(tmp_path / "blip.jsonl"),
                        use_cd=True, noise_step=500, cd_beta=0.1, max_length=4, cd_greedy=True, output_scores=True)
    lines = [json.loads(l) for l in open(tmp_path / "blip.jsonl")]
    assert [l["question_id"] for l in lines] == list(range(5))
    assert set(lines[0]) == {"question_id", "prompt", "text", "model_id", "image", "naive", "noise", "zeros", "metadata"}
    assert lines[0]["prompt"].endswith(QUESTION_SUFFIX) and lines[0]["model_id"] == "instruct_blip"
    assert set(res["scores"]) == {"string_match", "naive", "noise", "zeros"}
    # the zeros prior is deterministic: recompute it directly (front end on a zero image -> plain step-0 top-10)
    for q, a in zip(qs, lines):
        p = q["text"] + QUESTION_SUFFIX
        emb, _ = front.build(torch.zeros(1, 3, 56, 56, device=DEV), [tok_llm(p)], eng.w.t["embed"], qformer_text_ids=[tok_qf(p)])
        o = eng.generate(None, inputs_embeds=emb, max_length=1, min_length=1, eos_token_id=2, pad_token_id=2, n_top=10, temperature=1.0, top_k=50)
        want = C.label_dict_from_top(o.top_tok[0].tolist(), o.top_prob[0].tolist(), decode_token)
        pg, pw = np.array(C.get_prob_from_logits(a["zeros"])), np.array(C.get_prob_from_logits(want))
        assert np.abs(pg - pw).max() <= 0.02 + 0.05 * pw.max()
        assert a["noise"] != a["zeros"]
    assert map_pad_to_eos(torch.tensor([[5, 0, 0], [0, 7, 2]])).tolist() == [[5, 2, 2], [2, 7, 2]]
    # VCD at step 0 only (alpha 0.5 = the sampler default the reference driver leaves in place): masked entries appear in the
    # main pass's scores


def test_drivers_at_7b_widths_with_shared_image_prefixes(tmp_path):
    """run_pope and run_mme over an engine with LLaVA-1.5-7B widths (2 decoder layers): 48 POPE questions (8 images x 6: grouped prefix
    attention, captured graph, EOS) and 32 MME questions (2 per image), answers files + scorers + converter end to end."""
    from llava_align_amd import calibrate as C
    from llava_align_amd.engine import LlavaConfig, LlavaWeights, LMConfig, VddLlavaEngine, VisionConfig
    from llava_align_amd.mme_driver import llava_mme_inputs, run_mme
    from llava_align_amd.pope_driver import run_pope
    cfg = LlavaConfig(LMConfig(n_layers=2, max_pos=1024), VisionConfig(layers=3), "drivers-7b-widths")
    eng = VddLlavaEngine(cfg, weights=LlavaWeights.random(cfg, DEV, seed=5, std=0.02, lm_head_gain=2.0), device=DEV, use_graph=True)
    images = {}

    def load_image(name):
        if name not in images:
            images[name] = torch.randn(3, 336, 336, generator=torch.Generator().manual_seed(len(images)))
        return images[name]
    pope_q = [{"question_id": i, "image": f"im{i // 6}.jpg", "text": f"Is there a thing{i} in the image?", "label": ("yes", "no")[i % 2]} for i in range(48)]
    enc = lambda text, with_image: [t % 31990 + 3 if t >= 0 else t for t in toy_encode(("<image> " if with_image else "") + "system prompt of some length here . " + text)]
    res = run_pope(eng, pope_q, enc, decode, load_image, answers_path=str(tmp_path / "pope.jsonl"), batch_questions=48, unk_token_id=0, eos_token_id=2,
                   pad_token_id=0, max_new_tokens=6, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, seed=1)
    lines = [json.loads(l) for l in open(tmp_path / "pope.jsonl")]
    assert len(lines) == 48 and all(set(l["naive"]) and l["logits_score"] == C.get_prob_from_logits(l["naive"]) for l in lines)
    assert set(res["scores"]) == {"string_match", "naive", "none", "unk", "none_unk"}
    qs, gt = mme_questions(n_img=2)
    build = llava_mme_inputs(lambda p: [t % 31990 + 3 if t >= 0 else t for t in toy_encode(p)], load_image, unk_token_id=0)
    out = run_mme(eng, qs, build, decode, answers_path=str(tmp_path / "mme.jsonl"), batch_questions=32, max_new_tokens=4, eos_token_id=2, pad_token_id=0,
                  gt=gt, results_root=str(tmp_path / "res"), experiment="w7b", use_dd=True, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, seed=2)
    assert len(out["answers"]) == 32 and all(out["scores"][n] is not None for n in ("naive", "none", "unk", "none_unk"))


def test_run_pope_sharded_over_two_ranks_equals_one_rank(tmp_path):
    """SURVEY 8e / VERDICT r3 item 4: `torchrun -m ...pope_driver` shape - every rank runs run_pope on the same list, decodes its chunk
    of whole images, ONE gather, rank 0 writes the file.  Two ranks (sharing this box's one GPU over gloo) return, on BOTH ranks, the
    answers of the one-rank run, token for token under cd_greedy, and the JSONL is written once."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worker = os.path.join(root, "tests", "pope_shard_worker.py")
    env = dict(os.environ, VDD_FORCE_DEVICE="0", VDD_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    one = str(tmp_path / "one")
    p1 = subprocess.run([sys.executable, worker, one], capture_output=True, text=True, env=env, timeout=600)
    assert p1.returncode == 0, p1.stderr[-3000:]
    two = str(tmp_path / "two")
    p2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", "29547", worker, two], capture_output=True, text=True, env=env, timeout=900)
    assert p2.returncode == 0, p2.stderr[-3000:]
    r1 = json.load(open(one + ".rank0.json"))
    ra, rb = json.load(open(two + ".rank0.json")), json.load(open(two + ".rank1.json"))
    assert (r1["world"], ra["world"], rb["world"], ra["rank"], rb["rank"]) == (1, 2, 2, 0, 1)
    assert [a["question_id"] for a in r1["answers"]] == list(range(1000, 1021))
    for r in (ra, rb):                                        # every rank holds the full, identical result
        assert [a["text"] for a in r["answers"]] == [a["text"] for a in r1["answers"]]
        for x, y in zip(r["answers"], r1["answers"]):
            for name in ("naive", "none", "unk"):
                assert x[name].keys() == y[name].keys() and all(abs(x[name][k] - y[name][k]) <= 1e-6 for k in x[name]), (x["question_id"], name)
        assert r["scores"] == r1["scores"]
    lines = [json.loads(l) for l in open(two + ".jsonl")]
    assert [l["question_id"] for l in lines] == list(range(1000, 1021)) and lines == [json.loads(l) for l in open(one + ".jsonl")]


def test_run_mme_sweep_shares_one_prefill_per_batch_between_its_settings(eng, tmp_path):
    """run_llava.py:281-318 walks the question file once per sampling setting; run_mme(sweep=[...]) walks it once: the settings of a batch decode
    from one vision-tower pass and one prefill per pass type (engine.generate(reuse_prefill=True)).  Each setting's answers, label dicts and
    converted files equal those of a run of its own."""
    from llava_align_amd.mme_driver import llava_mme_inputs, run_mme
    qs, gt = mme_questions()
    images = {}

    def load_image(name):
        if name not in images:
            images[name] = torch.randn(3, 56, 56, generator=torch.Generator().manual_seed(len(images)))
        return images[name]
    build = llava_mme_inputs(toy_encode, load_image, unk_token_id=0)
    kw = dict(batch_questions=12, max_new_tokens=3, gt=gt, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, cd_greedy=True)
    settings = [dict(tag="default", temperature=1.0, top_p=None, top_k=None), dict(tag="temp_0.3", temperature=0.3, top_p=None, top_k=None),
                dict(tag="top_p_0.6", temperature=1.0, top_p=0.6, top_k=None)]
    for s_ in settings:
        s_["answers_path"] = str(tmp_path / "sweep" / f"ans-{s_['tag']}.jsonl")
    reused = []
    real = eng.generate
    eng.generate = lambda *a, **k: (lambda o: reused.append(bool(o.stats.get("prefill_reused"))) or o)(real(*a, **k))
    try:
        res = run_mme(eng, qs, build, decode, results_root=str(tmp_path / "res"), experiment="sw", sweep=settings, **kw)
    finally:
        eng.generate = real
    n_batches = 3                                              # 32 questions, 12 per batch
    assert len(reused) == n_batches * 3 * 3 and sum(reused) == n_batches * 3 * 2          # per batch and pass type: one prefill, two decodes from it
    assert set(res["runs"]) == {"default", "temp_0.3", "top_p_0.6"}
    for s_ in settings:
        alone = run_mme(eng, qs, build, decode, temperature=s_["temperature"], top_p=s_["top_p"], top_k=s_["top_k"], **kw)
        got = res["runs"][s_["tag"]]
        strip = lambda a: {k: v for k, v in a.items() if k != "answer_id"}
        assert [strip(a) for a in got["answers"]] == [strip(a) for a in alone["answers"]], s_["tag"]
        assert got["converted"] == alone["converted"]
        lines = [json.loads(l) for l in open(s_["answers_path"])]
        assert [strip(l) for l in lines] == [strip(a) for a in got["answers"]]
        assert sorted(os.listdir(got["results"]["naive"])) == sorted(c + ".txt" for c in CATS)
    assert res["runs"]["default"]["answers"] != res["runs"]["temp_0.3"]["answers"]            # (the label dicts depend on the temperature)
