"""Host-side logic of the drop-in (no GPU): warper absorption, criteria parsing, the
install hook, and the guarantee that the product path has no CPU fallback."""
import os
import re

import pytest
import torch
import transformers
from transformers.generation.logits_process import (LogitsProcessorList, MinLengthLogitsProcessor,
                                                    TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper,
                                                    TypicalLogitsWarper)

import llava_align_amd as L
from llava_align_amd import vcd_sample as VS
from llava_align_amd.sampling import thread_major_order

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_warpers_are_absorbed_in_hf_order():
    procs, spec, fused = VS._split_warpers(LogitsProcessorList(), LogitsProcessorList(
        [TemperatureLogitsWarper(0.2), TopKLogitsWarper(50), TopPLogitsWarper(0.9)]))
    assert fused and procs == [] and (spec.temperature, spec.top_k, spec.top_p) == (0.2, 50, 0.9)
    # merged list (transformers >= 4.39 `_sample`)
    procs, spec, fused = VS._split_warpers(LogitsProcessorList([TemperatureLogitsWarper(0.7), TopKLogitsWarper(3)]), None)
    assert fused and (spec.temperature, spec.top_k, spec.top_p) == (0.7, 3, None)
    # a real processor in front keeps the warpers fused but the step split in two launches
    mlp = MinLengthLogitsProcessor(5, eos_token_id=2)
    procs, spec, fused = VS._split_warpers(LogitsProcessorList([mlp]), LogitsProcessorList([TopKLogitsWarper(1)]))
    assert not fused and procs == [mlp] and spec.top_k == 1
    # unknown or re-ordered warpers run in Python as a whole
    procs, spec, fused = VS._split_warpers(None, LogitsProcessorList([TopPLogitsWarper(0.9), TopKLogitsWarper(5)]))
    assert not fused and len(procs) == 2 and spec.top_k is None
    procs, spec, fused = VS._split_warpers(None, LogitsProcessorList([TypicalLogitsWarper(0.5)]))
    assert not fused and len(procs) == 1
    # non -inf filter value cannot be fused
    procs, spec, fused = VS._split_warpers(None, LogitsProcessorList([TopKLogitsWarper(5, filter_value=-1e4)]))
    assert not fused


def test_warpspec_activation_rules():
    w = L.WarpSpec(temperature=1.0, top_k=0, top_p=1.0)
    assert (w.t, w.k, w.p) == (0.0, 0, 2.0)          # all three off, as HF skips them
    w = L.WarpSpec(temperature=0.2, top_k=1, top_p=0.0)
    assert (w.t, w.k, w.p) == (0.2, 1, 0.0)


def test_criteria_parsing():
    sc = transformers.StoppingCriteriaList([transformers.MaxLengthCriteria(max_length=77)])
    rest, ml, eos = VS._parse_criteria(sc, None)
    assert rest == [] and ml == 77 and eos is None
    if hasattr(transformers, "EosTokenCriteria"):
        sc.append(transformers.EosTokenCriteria(eos_token_id=[2, 5]))
        rest, ml, eos = VS._parse_criteria(sc, 50)
        assert ml == 50 and torch.as_tensor(eos).tolist() == [2, 5]


def test_thread_major_order_is_a_permutation():
    for V, dt in ((97, torch.float16), (1003, torch.float32), (20000, torch.bfloat16)):
        o = thread_major_order(V, dt)
        assert sorted(o) == list(range(V))
    o = thread_major_order(20000, torch.float16)
    from llava_align_amd.sampling import KERNEL_THREADS
    assert o[:8] == list(range(8)) and o[8:16] == list(range(8 * KERNEL_THREADS, 8 * KERNEL_THREADS + 8))   # thread 0: chunk 0, then chunk KERNEL_THREADS


def test_install_hook_is_idempotent():
    mixin = transformers.generation.utils.GenerationMixin
    saved = (mixin.__dict__.get("sample"), mixin.__dict__.get("_sample"))
    try:
        L.evolve_vcd_sampling()
        L.evolve_vcd_sampling()
        assert mixin.sample is VS.sample
        if saved[1] is not None:
            assert mixin._sample is VS._sample_v5
    finally:
        if saved[0] is None:
            del mixin.sample
        else:
            mixin.sample = saved[0]
        if saved[1] is not None:
            mixin._sample = saved[1]


def test_no_cpu_fallback():
    v = torch.zeros(2, 97, dtype=torch.float16)
    with pytest.raises(L.VddLibraryError):
        L.contrast_sample(v, v)
    if not torch.cuda.is_available():
        with pytest.raises(L.VddLibraryError):
            L.add_diffusion_noise(torch.zeros(3, 4, 4), 500)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from llava_align_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("VDD_HIP_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(L.VddLibraryError, match="no CPU fallback"):
        _lib.load_lib()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "llava-align_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "vdd_oracle" not in src, f


def test_eos_without_pad_raises_like_the_reference():
    class M:
        generation_config = transformers.GenerationConfig()
    ids = torch.ones(1, 4, dtype=torch.long)
    with pytest.raises(ValueError, match="make sure that `pad_token_id` is defined"):
        VS.sample(M(), ids, eos_token_id=2, pad_token_id=None)


def test_driver_prompt_builders_follow_the_reference_format_strings():
    """CPU-only host logic of the MME / InstructBLIP drivers: the prompt strings the reference builds (run_llava.py:52-62,101-115,
    run_qwen.py:101-104,176-177, blip_calibrate.py:42,74; conversation.py:252-262) and the token 0 -> 2 post-map
    (blip2_vicuna_instruct.py:414)."""
    import torch
    from llava_align_amd.blip_driver import QUESTION_SUFFIX, map_pad_to_eos
    from llava_align_amd.mme_driver import MME_SUBSETS, ONE_WORD, llava_mme_inputs, qwen_mme_inputs, vicuna_v1_prompt
    assert vicuna_v1_prompt("hi") == ("A chat between a curious user and an artificial intelligence assistant. The assistant gives helpful, "
                                      "detailed, and polite answers to the user's questions. USER: hi ASSISTANT:")
    assert ONE_WORD == QUESTION_SUFFIX == " Please answer this question with one word."
    assert len(MME_SUBSETS) == 8 and "existence" in MME_SUBSETS and "artwork" not in MME_SUBSETS
    seen = []

    def encode(prompt):
        seen.append(prompt)
        return [1] + [(-200 if w == "<image>" else 7) for w in prompt.replace("<image>", " <image> ").split()]
    loads = []
    build = llava_mme_inputs(encode, lambda name: loads.append(name) or torch.zeros(3, 2, 2), unk_token_id=0)
    line = {"image": "a.png", "text": "Is it red?"}
    main, none, unk = build(line, "main"), build(line, "none"), build(line, "unk")
    assert seen[0].endswith("USER: <image>\nIs it red? ASSISTANT:")                                   # main: no one-word suffix
    assert seen[1].endswith("USER: Is it red? Please answer this question with one word. ASSISTANT:")    # none: no image token
    assert seen[2].endswith("USER: <image>\nIs it red? Please answer this question with one word. ASSISTANT:")
    assert (main["input_ids"] == -200).sum() == 1 and main["image"] is not None
    assert (none["input_ids"] == -200).sum() == 0 and none["image"] is None
    assert (unk["input_ids"] == -200).sum() == 0 and (unk["input_ids"] == 0).sum() == 1 and unk["image"] is None
    build(line, "main")
    assert loads == ["a.png"]                                                                          # decoded once, cached
    texts = []
    qb = qwen_mme_inputs(lambda text, path: texts.append((text, path)) or torch.zeros(1, 4), image_path=lambda f: "/imgs/" + f)
    for kind in ("main", "none", "unk"):
        qb(line, kind)
    assert texts == [("<img>/imgs/a.png</img>Is it red? Answer:", "/imgs/a.png"), ("Is it red? Answer:", None), ("None Is it red? Answer:", None)]
    assert map_pad_to_eos(torch.tensor([[5, 0, 0], [0, 7, 2]])).tolist() == [[5, 2, 2], [2, 7, 2]]


def test_qwen_pope_driver_passes_prompts_and_kwargs_like_the_reference(monkeypatch):
    """qwen_driver.run_qwen_pope against a recording engine (host logic only): the five passes' prompt strings (qwen_calibrate.py:36-41, :97),
    which image each pass hands the front-end, the generate kwargs of the main pass vs the plain prior passes (:43-65, :113-136), image-span
    sharing keys, the file's fields."""
    import json
    import types
    import torch
    from llava_align_amd import vcd_add_noise
    from llava_align_amd.qwen_driver import PRIORS, run_qwen_pope
    calls, fronts = [], []
    monkeypatch.setattr(vcd_add_noise, "add_diffusion_noise", lambda im, t: im + torch.randn_like(im) * (1 + t))     # (the product's is a GPU kernel)

    class Engine:
        device = torch.device("cpu")

        def clear_image_cache(self):
            pass

        def generate(self, ids, inputs_embeds=None, **kw):
            calls.append(dict(kw, n=len(inputs_embeds)))
            n, T = len(inputs_embeds), kw["max_new_tokens"]
            return types.SimpleNamespace(tokens=torch.full((n, T), 5), top_tok=torch.arange(10).repeat(n, 1), top_prob=torch.full((n, 10), 0.1))

    def embed_prompt(text, image):
        fronts.append((text, None if image is None else ("zero" if not bool(image.any()) else round(float(image.std()), 1))))
        e = torch.zeros(6, 4)
        return e if image is None else (e, 4)
    imgs = {"a.jpg": torch.ones(3, 4, 4) * torch.arange(4.0), "b.jpg": torch.ones(3, 4, 4) * torch.arange(4.0) * 3}
    qs = [{"question_id": i, "image": "ab"[i // 2] + ".jpg", "text": f"Is it {i}?", "label": "yes"} for i in range(4)]
    torch.manual_seed(0)
    out = run_qwen_pope(Engine(), qs, embed_prompt, lambda ids: " ".join(map(str, ids)), lambda n: imgs[n], image_path=lambda f: "/d/" + f,
                        use_cd=True, noise_step=500, use_dd=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, top_k=None, seed=7)
    assert [c["n"] for c in calls] == [4] * 5 and len(fronts) == 4 * 6
    main, none, unk, noise, zero = calls
    assert main["max_new_tokens"] == 20 and main["min_new_tokens"] == 1 and main["eos_token_id"] == main["pad_token_id"] == 151643
    assert main["use_dd"] and len(main["images_cd"]) == 4 and main["n_top"] == 10 and main["seed"] == 7
    assert [k[0] for k, _ in main["embeds_prefix"]] == ["clean"] * 4 and main["embeds_prefix"][0] == main["embeds_prefix"][1] != main["embeds_prefix"][2]
    for c in (none, unk, noise, zero):                           # plain sampling: no VDD / VCD kwargs, one token, EOS floor as in the main call
        assert "use_dd" not in c and "images_cd" not in c and c["max_new_tokens"] == 1 and c["min_new_tokens"] == 1 and c["temperature"] == 0.2
    assert "embeds_prefix" not in none and "embeds_prefix" not in noise and len({k for k, _ in zero["embeds_prefix"]}) == 1
    cd, mn, nn_, un, ns, zr = (fronts[i * 4:(i + 1) * 4] for i in range(6))
    assert mn[1] == ("<img>/d/a.jpg</img>Is it 1? Answer:", 1.1) and mn[2][1] == 3.4
    assert nn_[3] == ("Is it 3? Answer:", None) and un[0] == ("None Is it 0? Answer:", None)
    assert [t for t, _ in ns] == [t for t, _ in mn] == [t for t, _ in zr] == [t for t, _ in cd] and all(v == "zero" for _, v in zr)
    assert all(v not in (1.1, 3.4, "zero") for _, v in ns + cd)                            # noised copies, drawn per question
    assert PRIORS == ("none", "unk", "noise", "zero")
    a = out["answers"][2]
    assert list(a) == ["question_id", "prompt", "text", "naive", "noise", "none", "zero", "unk", "model_id", "image", "metadata"]
    assert a["prompt"] == "<img>/d/b.jpg</img>Is it 2? Answer:" and a["text"] == " ".join(["5"] * 20) and out["world"] == 1


def test_norm_fused_row_limit_follows_the_lds_budget():
    """ops.norm_fused_rows: rows whose normalised copy (2 d bytes each) fits the 142 KiB the normalise-once projections may use; widths the
    kernel's chunk map does not cover (d % 256, d > 8192) take the unfused layer."""
    from llava_align_amd import ops
    assert [ops.norm_fused_rows(d) for d in (4096, 5120, 8192, 2048, 16384, 4000)] == [16, 14, 8, 16, 0, 0]


def test_projection_and_layer_forms_are_measured_choices_with_generic_fallbacks(monkeypatch):
    """Round 6: which FORM a projection / a few-row layer takes is a measured, persisted choice (ops._pick_form, ops.norm_fused_pays), not a
    table of crossovers measured on three model shapes.  Host logic only: recorded choices win, eligibility limits hold, and without a
    measurement (no GPU here: nothing can be timed) the shape-generic fallbacks apply."""
    from llava_align_amd import ops
    monkeypatch.setattr(ops, "_form_choice", {})
    assert ops._skinny_serves(16, 4096) and ops._skinny_serves(64, 4096) and not ops._skinny_serves(65, 4096)
    assert ops._skinny_serves(16, 4224) and not ops._skinny_serves(17, 4224)                      # 17 - 64 rows need K % 256
    assert ops._form_key("linear", 40, 12288, 4096, 2) == ("linear", 40, 12288, 4096, 2)          # exact rows up to 64 ...
    assert ops._form_key("to_norm", 100, 4096, 11008, 2)[1] == 128                                # ... 64-row buckets above
    # fallbacks: weight-streaming up to 16 rows, the norm-fused layer up to 8 (and never where its kernels do not exist)
    assert ops.skinny_rows(12288, 4096) == ops.FALLBACK_SKINNY_ROWS == 16
    assert [m for m in range(1, 20) if ops.norm_fused_pays(m, 4096)] == list(range(1, ops.FALLBACK_FUSED_ROWS + 1))
    assert not ops.norm_fused_pays(2, 4000) and not ops.norm_fused_pays(15, 5120)                 # no such kernel / beyond its LDS image (14 rows at d = 5120)
    # recorded measurements win
    ops.gemm_choices_import({"form,layer,10,4096,0,2": "plain", "form,layer,13,4096,0,2": "fused", "form,linear,3,32000,5120,2": "gemm",
                             "form,linear,1,32000,5120,2": "skinny", "form,linear,2,32000,5120,2": "skinny"})
    assert not ops.norm_fused_pays(10, 4096) and ops.norm_fused_pays(13, 4096) and ops.norm_fused_pays(13, 4096, __import__("torch").bfloat16)
    assert ops.skinny_rows(32000, 5120) == 2                                                       # (LLaVA-1.5-13B's lm_head leaves the weight-streaming kernel early)
    assert ops.form_choices_export()["form,layer,10,4096,0,2"] == "plain"
    # overrides for tests / probes
    monkeypatch.setattr(ops, "FORCE_LAYER_FORM", "fused")
    assert ops.norm_fused_pays(10, 4096) and not ops.norm_fused_pays(17, 4096)
    with ops.batch_invariant():
        assert not ops.norm_fused_pays(2, 4096) and ops.skinny_rows(32000, 5120) == 0


def test_one_launch_attention_band_and_its_batch_invariant_form():
    """Ungrouped decode steps take the one-launch RoPE + KV write + attention kernel up to 32 rows (round 5: tools/few_row_curve.py).  In
    batch-invariant mode every op has ONE form (round 6): no one-launch attention, no weight-streaming projections, no norm-fused layer,
    no split-K slabs - whatever the row count - and `with ops.batch_invariant()` scopes the mode."""
    from llava_align_amd import ops
    assert ops.FUSED_ATTN_MAX_M == 16 and ops.FUSED_ATTN_UNGROUPED_MAX_M == 32 and ops.fused_attention_rows() == 32
    assert not ops.GEMM_BATCH_INVARIANT
    with ops.batch_invariant():
        assert ops.GEMM_BATCH_INVARIANT and ops.fused_attention_rows() == 0
        assert ops.skinny_rows(4096, 4096) == ops.skinny_rows(32000, 4096) == ops.skinny_rows(27648, 5120) == 0
        assert not any(ops.norm_fused_pays(m, 4096) for m in range(1, 20))
        with ops.batch_invariant(False):
            assert ops.fused_attention_rows() == 32
        assert ops.GEMM_BATCH_INVARIANT
    assert not ops.GEMM_BATCH_INVARIANT and ops.skinny_rows(4096, 4096) > 0


def test_drivers_select_batch_invariance_for_deterministic_decodes():
    from llava_align_amd.shard import resolve_batch_invariant as r
    assert r(None, 1, dict(cd_greedy=True)) and r(None, 8, dict(top_k=1)) and r(None, 1, dict(do_sample=False))
    assert not r(None, 8, dict(temperature=0.2, seed=1)) and not r(None, 1, dict(top_k=50)) and not r(None, 1, {})
    assert r(True, 1, {}) and not r(False, 8, dict(cd_greedy=True))


def test_in_tree_gemm_defaults_are_bound_to_the_kernel_source():
    """gemm_choices_mi355x.json carries the hash of the GEMM kernel SOURCE it was measured on (ops.gemm_source_fingerprint);
    a kernel edit without re-measuring (or re-stamping) the defaults fails here instead of silently shipping stale picks."""
    import json
    import os
    from llava_align_amd import ops
    pkg = os.path.dirname(os.path.abspath(ops.__file__))
    doc = json.load(open(os.path.join(pkg, "gemm_choices_mi355x.json")))
    assert doc["gemm_source_sha"] == ops.gemm_source_fingerprint() != ""


def test_tuner_cache_merges_under_the_lock(tmp_path, monkeypatch):
    """Two writers (ranks of one node) append different shapes: read - merge - replace under the advisory lock keeps both."""
    import json
    from llava_align_amd import ops
    path = str(tmp_path / "choices.json")
    monkeypatch.setitem(ops._persist, "path", path)
    monkeypatch.setitem(ops._persist, "section", "dev|abc")
    saved = dict(ops._gemm_choice)
    try:
        with ops._CacheLock():
            ops._store_persisted((1, 4096, 4096, 0, False, 2), 40)
        with ops._CacheLock():
            ops._store_persisted((2, 4096, 4096, 0, False, 2), 42)
        doc = json.load(open(path))
        assert doc["dev|abc"] == {"1,4096,4096,0,False,2": 40, "2,4096,4096,0,False,2": 42}
        ops._gemm_choice.clear()
        ops._read_cache_section()
        assert ops._gemm_choice[(1, 4096, 4096, 0, False, 2)] == 40 and ops._gemm_choice[(2, 4096, 4096, 0, False, 2)] == 42
    finally:
        ops._gemm_choice.clear()
        ops._gemm_choice.update(saved)


def test_cache_lock_is_reentrant_within_a_process(tmp_path, monkeypatch):
    """Measuring a projection FORM holds the cache lock and runs the GEMM, whose tile tuner takes it again: a second flock on another
    descriptor of the same file would wait forever (it did, on the GPU box, for 20 minutes)."""
    from llava_align_amd import ops
    monkeypatch.setitem(ops._persist, "path", str(tmp_path / "choices.json"))
    monkeypatch.setitem(ops._persist, "section", "dev|lib")
    with ops._CacheLock():
        with ops._CacheLock():
            ops._store_persisted(("form", "linear", 3, 32000, 5120, 2), "gemm")
            ops._store_persisted((1, 4096, 4096, 0, False, 2), 17)
        assert ops._CacheLock._depth == 1 and ops._CacheLock._file is not None
    assert ops._CacheLock._depth == 0 and ops._CacheLock._file is None
    import json
    sec = json.load(open(tmp_path / "choices.json"))["dev|lib"]
    assert sec == {"form,linear,3,32000,5120,2": "gemm", "1,4096,4096,0,False,2": 17}
    monkeypatch.setattr(ops, "_form_choice", {})
    monkeypatch.setattr(ops, "_gemm_choice", {})
    ops._read_cache_section()
    assert ops._form_choice == {("linear", 3, 32000, 5120, 2): "gemm"} and ops._gemm_choice == {(1, 4096, 4096, 0, False, 2): 17}


def test_gemm_config_word_and_form_defaults():
    """vdd_gemm's `config`: tile id in bits 0-3 and 6-7 (ids 16.. since round 6), schedule in bits 4-5 - ids up to 15 stay `tile + 16 * sched`;
    the tuner has a bucket of its own for <= 32 rows (the 32 x 128 tiles serve only those); the in-tree FORM defaults are bound to the kernel sources."""
    import json
    import os
    from llava_align_amd import ops
    assert [ops.gemm_config(c, s) for c, s in ((1, 0), (9, 2), (15, 1))] == [1, 9 + 32, 15 + 16]
    assert ops.gemm_config(16, 0) == 64 and ops.gemm_config(16, 2) == 64 + 32 and ops.gemm_config(17, 1) == 1 + 16 + 64
    dec = lambda v: ((v & 15) | (((v >> 6) & 3) << 4), (v >> 4) & 3)
    assert all(dec(ops.gemm_config(c, s)) == (c, s) for c in range(1, 20) for s in range(3))
    assert ops._gemm_key(2, 4096, 4096, 0)[0] == ops._gemm_key(32, 4096, 4096, 0)[0] == -1
    assert ops._gemm_key(33, 4096, 4096, 0)[0] == ops._gemm_key(64, 4096, 4096, 0)[0] == 1 and ops._gemm_key(65, 4096, 4096, 0)[0] == 2
    pkg = os.path.dirname(os.path.abspath(ops.__file__))
    doc = json.load(open(os.path.join(pkg, "form_choices_mi355x.json")))
    assert doc["kernel_source_sha"] == ops.kernel_source_fingerprint() != "" and len(doc["choices"]) > 1000
    assert all(k.startswith("form,") and v in ("skinny", "gemm", "slabs", "fused", "plain") for k, v in doc["choices"].items())
    g = json.load(open(os.path.join(pkg, "gemm_choices_mi355x.json")))["choices"]
    small = [dec(v)[0] for k, v in g.items() if k.split(",")[0] in ("-1", "1") and k.split(",")[1:3] in (["12288", "4096"], ["4096", "4096"], ["4096", "11008"])]
    assert small and all(t in (12, 13, 14, 15) for t in small)                 # the few-dozen-row tiles of round 6 carry the 7B projections up to 64 rows
