"""bench.py — python bench.py --gpus N --steps K --warmup W  -> ONE JSON line on rank 0.

Metric (BASELINE.json): decode tokens/sec under VDD dual-pass, LLaVA-1.5-7B, POPE.
Workload (synthetic — no checkpoints / tokenizer / images exist on either box): LLaVA-1.5-7B shapes with
N(0, 0.02) bf16 weights (lm_head x4 so that the contrast keeps a handful of candidates per row, as a trained model does on POPE);
POPE-like prompts = 35 system tokens + 1 image slot (576 patch embeddings) + 19..28 question tokens, 6 questions per 336x336
image; use_dd_unk, cd_alpha=1, cd_beta=0.1, T=0.2, 64 new tokens, no EOS (steady-state variant).

A "step" = one generate() over one batch of `--questions` questions: ViT + projector per distinct image, prefill of both
branches, 64 decode steps with the fused contrastive tail, step-0 top-10 probabilities.  Everything is inside the timed region.
value = generated tokens / wall time over all ranks (weak scaling: every rank runs its own batch of the same size); the per-question
results {qid, n_tokens, tokens, top10_tok, top10_prob} are gathered once at the end with RCCL (shard.gather_results).

`--gpus N` with N > 1 and no torchrun environment re-launches itself as N ranks (python -m torch.distributed.run on 127.0.0.1);
under torchrun (the driver's launch) it reads RANK / LOCAL_RANK / WORLD_SIZE.  n_gpus on the line is the real world size.

Extra objects on the line:
  roofline      the fused contrastive sampling kernel (the kernel north_star prices) at B=4096 rows, HIP events on the launch
                stream: frac = SURVEY §8d algorithmic bytes / time / 8 TB/s, frac_traffic = the bytes the launch really moves
                (computed from the survivor count of the same inputs: c is only read where a candidate survives the beta-mask)
  roofline_extra  the regimes where nothing is masked (beta = 1e-6) and V = 151,936
  pope_eos      the same batch stopped by EOS after 1-2 tokens per question (POPE answers), EOS logic of the kernel in the loop
  decode_step   measured ms per decode step of the engine vs the weight-streaming floor
  cpu_baseline  the reference path on the host CPU: value = the headline's own model (LLaVA-1.5-7B shapes, one question, B = 1, 2 new
                tokens, all 32 layers measured); `config1` = BASELINE config #1 end to end on the toy LM; `sampling_tail` = the per-step tail with a
                stubbed forward (ms/step, comparable with BASELINE.md §2)
  config2_full / config3 / config4 / config5   the other BASELINE configs through the drivers (pope_driver / mme_driver / blip_driver) and config #3's
                13B shapes on this GPU and at its rank-of-8 share, each with a decode-step HBM roofline (`step_roofline`) from HIP events
  batch_invariant   the headline step in batch-invariant mode (what the drivers select for deterministic decodes)
  llava_bench_eos   config #3's call shape with EOS: static / retirement, and a 360-question list through generate_list against batch-after-batch
  eager_gpu     the reference path on this GPU: the oracle restatement of the patched sample() over HF's OWN eager stack - the installed
                transformers' LlamaForCausalLM + CLIPVisionModel composed like LlavaLlamaForCausalLM (tests/hf_llava.py), 7B widths, 32 + 24
                layers, fp16 (builder.py:40), eager attention with output_attentions=True (llava_calibrate.py:175) - B=1, one forward
                per branch per token; `dropin_gpu` = this package's sample() over the same object
  eager_gpu_port / dropin_gpu_port   the same two over tests/ref_llava.py (this repo's plain-torch LLaVA, bf16): the comparator of
                rounds 1 - 4, kept for continuity
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
N_NEW = 64


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--questions", type=int, default=768, help="questions per generate() batch per GPU (6 per image); 768 = 1,536 decode rows, ~190 GB of KV pools + weights")
    ap.add_argument("--model", default="llava-1.5-7b")
    ap.add_argument("--no-baselines", action="store_true")
    ap.add_argument("--strong", type=int, default=0, metavar="N_QUESTIONS",
                    help="strong-scaling mode: ONE seeded list of N questions (6 per image) split over the ranks by shard.get_chunk "
                         "(the reference's contiguous ceil-chunks, whole image groups), EOS on, ragged results gathered once")
    return ap.parse_args()


def maybe_relaunch(a):
    """`python bench.py --gpus N` (N > 1) outside torchrun: become N ranks on this node."""
    if a.gpus <= 1 or "RANK" in os.environ:
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def pope_prompts(n_img, per_img=6, seed=1234, vocab=32000, n_sys=35, txt=(19, 29), image=336):
    """[35 system tokens] + [-200] + [19..28 question tokens], 6 questions per synthetic image."""
    import numpy as np
    import torch
    rng = np.random.default_rng(seed)
    sys_tok = [1] + rng.integers(3, vocab, size=n_sys - 1).tolist()
    ids, imgs = [], []
    g = torch.Generator().manual_seed(7 + seed)
    for _ in range(n_img):
        im = torch.randn(3, image, image, generator=g)
        for _ in range(per_img):
            t = rng.integers(3, vocab, size=int(rng.integers(*txt))).tolist()
            ids.append(torch.tensor(sys_tok + [-200] + t))
            imgs.append(im)
    return ids, imgs


# ------------------------------------------------------------------ fused-kernel roofline leg
def kernel_point(dev, B, V, beta=0.1, n_in=2, scores=True, iters=100, warmup=10):
    """One launch shape of vdd_contrast_sample: logits N(0, 4) with a planted row maximum, use_dd_unk, T = 0.2."""
    import torch
    import llava_align_amd as L
    dtype = torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(0)
    v = (torch.randn(B, V, device=dev, generator=g) * 4).to(dtype)
    v[torch.arange(B, device=dev), torch.randint(0, V, (B,), device=dev, generator=g)] = 25.0   # planted row max
    c = (v.float() + torch.randn(B, V, device=dev, generator=g) * 1.5).to(dtype)
    out_scores = torch.empty(B, V, dtype=dtype, device=dev) if scores else None
    toks = torch.empty(B, dtype=torch.long, device=dev)
    spec = L.WarpSpec(temperature=0.2)
    run = lambda i: L.contrast_sample(v, c, None, alpha=1.0, beta=beta, warp=spec, out_tokens=toks, out_scores=out_scores, seed=0, offset=i)
    for i in range(warmup):
        run(i)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # HIP events on the launch stream
    e0.record()
    for i in range(iters):
        run(i)
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / iters
    es = 2
    alg = B * ((n_in + (1 if scores else 0)) * V * es + 8)          # (n_in+n_out)*V*e + 8 per row
    # what the launch really moves: v once, the scores row once, and c only in the 16-byte chunks that hold a survivor of the
    # plausibility mask (masked entries are -inf whatever c holds) - counted on these very inputs
    cutoff = (v.float().max(-1, keepdim=True).values + torch.log(torch.tensor(beta)).item()).to(dtype)
    surv = v >= cutoff
    pad = (-V) % 8
    chunks = torch.nn.functional.pad(surv, (0, pad)).view(B, -1, 8).any(-1).sum().item()
    n_surv = surv.sum().item() / B
    traffic = B * V * es * (1 + (1 if scores else 0)) + chunks * 16 * (n_in - 1) + B * 8
    gbs, gbs_t = alg / (ms * 1e-3) / 1e9, traffic / (ms * 1e-3) / 1e9
    del v, c, out_scores, surv
    torch.cuda.empty_cache()
    # frac_traffic (the bytes the launch really moves) first; frac = the SURVEY 8d algorithmic-bytes fraction the contract defines
    return {"bound": "hbm", "frac_traffic": round(gbs_t / HBM_PEAK_GBS, 4), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": int(traffic), "traffic_source": "computed: v + scores rows + the c chunks holding a beta-mask survivor (counted on the "
            "bench inputs); rocprofv3 FETCH/WRITE cross-check in profiles/",
            "frac_algorithmic": round(gbs / HBM_PEAK_GBS, 4), "achieved_traffic_GBs": round(gbs_t, 1),
            "kernel": "vdd_contrast_sample_kernel<bf16, lds-row>" if V <= 86016 else "vdd_contrast_sample_kernel<bf16, global-row>",
            "launch_us": round(ms * 1e3, 2), "algorithmic_bytes_per_launch": alg, "survivors_per_row": round(n_surv, 2),
            "shape": {"B": B, "V": V, "dtype": "bf16", "n_in": n_in, "scores_out": scores, "beta": beta, "note": "use_dd_unk, T=0.2"}}


def pmc_traffic():
    """HBM bytes per launch of the fused kernel at the roofline shape from the latest committed rocprofv3 PMC passes
    (profiles/r*_pmc_fused_kernel.txt: FETCH_SIZE and WRITE_SIZE in KB, separate passes; gfx950 reads are counted at half: x 2) -
    the measured cross-check of the computed `traffic`; None when no profile file travels with the tree."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*pmc_fused_kernel.txt")))
    if not files:
        return None
    txt = open(files[-1]).read()
    f, w = re.search(r"FETCH_SIZE\s+n=\d+ mean ([0-9.]+) KB", txt), re.search(r"WRITE_SIZE\s+n=\d+ mean ([0-9.]+) KB", txt)
    if not (f and w):
        return None
    return {"bytes": int((2 * float(f.group(1)) + float(w.group(1))) * 1000), "source": os.path.relpath(files[-1], ROOT),
            "formula": "2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes)"}


# ------------------------------------------------------------------ reference-path baselines (oracle loop + eager torch model)
def reference_path(weights, device, ids, img, n_new, dtype=None, layers=None):
    """Runs the reference's decoding path for ONE question: oracle restatement of sample() (B=1, one forward per
    branch per token) over an eager torch LLaVA.  Returns seconds."""
    import torch
    from oracle import vdd_oracle as O
    from ref_llava import RefLlava
    model = RefLlava(weights, device=device, dtype=dtype or torch.bfloat16, output_attentions=True)
    kw = dict(images=img[None], attention_mask=torch.ones(1, ids.numel(), dtype=torch.long), use_cache=True,
              cd_alpha=1.0, cd_beta=0.1, use_dd_unk=True)
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    O.reference_loop(model, ids[None].clone(), warp=O.WarpConfig(temperature=0.2), max_length=ids.numel() + n_new,
                     pad_token_id=None, eos_token_id=None, pick=O.pick_multinomial, **kw)
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()
    return time.perf_counter() - t0


def dropin_path(weights, device, ids, img, n_new):
    """The SAME eager torch model driven by this package's drop-in `sample()` (INTEGRATION.md option 1: only the import
    changes in the reference's scripts): forwards untouched, the per-step tail replaced by the fused HIP kernel."""
    import torch
    import transformers
    from llava_align_amd import sample
    from ref_llava import RefLlava
    model = RefLlava(weights, device=device, dtype=torch.bfloat16, output_attentions=True, logits_on_device=True)
    ids_d = ids[None].to(device)
    kw = dict(images=img[None], attention_mask=torch.ones(1, ids.numel(), dtype=torch.long, device=device), use_cache=True,
              cd_alpha=1.0, cd_beta=0.1, use_dd_unk=True)
    crit = transformers.StoppingCriteriaList([transformers.MaxLengthCriteria(max_length=ids.numel() + n_new)])
    warp = transformers.LogitsProcessorList([transformers.TemperatureLogitsWarper(0.2)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sample(model, ids_d, logits_warper=warp, stopping_criteria=crit, **kw)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def bench_dropin_gpu(eng, dev, n_q=2, n_new=N_NEW):
    ids, imgs = pope_prompts(1, per_img=n_q, seed=99)
    dropin_path(eng.w, dev, ids[0], imgs[0], 2)
    dt = sum(dropin_path(eng.w, dev, ids[q], imgs[q], n_new) for q in range(n_q))
    return {"value": round(n_q * n_new / dt, 2), "unit": "tokens/s",
            "what": "drop-in sample() of this package (fused HIP tail, no per-step host sync) over the SAME eager bf16 torch model "
                    "as eager_gpu: what changing only the import in the reference's scripts buys; the engine is what removes "
                    "the per-branch forwards",
            "sample": f"{n_q} questions x {n_new} new tokens in {dt:.1f}s"}


def bench_eager_gpu(eng, dev, n_q=2, n_new=N_NEW):
    ids, imgs = pope_prompts(1, per_img=n_q, seed=99)
    reference_path(eng.w, dev, ids[0], imgs[0], 2)                      # warm-up (library handles)
    dt = sum(reference_path(eng.w, dev, ids[q], imgs[q], n_new) for q in range(n_q))
    return {"value": round(n_q * n_new / dt, 2), "unit": "tokens/s", "kind": "port",
            "what": "oracle restatement of the monkey-patched sample() over an eager bf16 torch-ROCm LLaVA-1.5-7B (tests/ref_llava.py: "
                    "this repo's own plain-torch model, checked against HF Llama / CLIP modules on CPU, not HF Llama itself): B=1, "
                    "one forward per branch per token, KV grown by torch.cat, attention maps materialised",
            "sample": f"{n_q} questions x {n_new} new tokens in {dt:.1f}s"}


def _hf_reference_stack(dev):
    """What `load_pretrained_model` hands the reference's scripts, minus the checkpoint (tests/hf_llava.py): the INSTALLED transformers'
    LlamaForCausalLM + CLIPVisionModel composed like LlavaLlamaForCausalLM, LLaVA-1.5-7B widths, 32 + 24 layers, fp16 (builder.py:40),
    eager attention (the reference era's LlamaAttention materialised its maps; llava_calibrate.py:175 asks for them)."""
    import torch
    import hf_llava
    return hf_llava.build(dev, torch.float16, d=4096, layers=32, heads=32, ffn=11008, vocab=32000, clip_width=1024, clip_layers=24, clip_heads=16,
                          clip_mlp=4096, image=336, patch=14, max_pos=4096, lm_head_gain=4.0, attn_implementation="eager")


def bench_hf_gpu(dev, n_q=2, n_new=N_NEW):
    """`eager_gpu` / `dropin_gpu` on HF's own eager stack in the reference's dtype: (a) the oracle restatement of the
    reference's patched sample() - B = 1, one forward per branch per token, attention maps materialised, tail on the logits as the loop
    reads them; (b) this package's drop-in sample() over the same object (fused HIP tail, logits stay on the device)."""
    import torch
    import transformers
    import hf_llava
    from oracle import vdd_oracle as O
    from llava_align_amd import sample
    m = _hf_reference_stack(dev)
    ids, imgs = pope_prompts(1, per_img=n_q, seed=99)

    def eager(q, n):
        proto = hf_llava.HfProto(m, output_attentions=True)
        kw = dict(images=imgs[q][None], attention_mask=torch.ones(1, ids[q].numel(), dtype=torch.long), use_cache=True, cd_alpha=1.0, cd_beta=0.1,
                  use_dd_unk=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        O.reference_loop(proto, ids[q][None].clone(), warp=O.WarpConfig(temperature=0.2), max_length=ids[q].numel() + n, pad_token_id=None,
                         eos_token_id=None, pick=O.pick_multinomial, **kw)
        torch.cuda.synchronize(); return time.perf_counter() - t0

    def dropin(q, n):
        proto = hf_llava.HfProto(m, output_attentions=True, logits_on_device=True)
        ids_d = ids[q][None].to(dev)
        kw = dict(images=imgs[q][None], attention_mask=torch.ones(1, ids[q].numel(), dtype=torch.long, device=dev), use_cache=True, cd_alpha=1.0,
                  cd_beta=0.1, use_dd_unk=True)
        crit = transformers.StoppingCriteriaList([transformers.MaxLengthCriteria(max_length=ids[q].numel() + n)])
        warp = transformers.LogitsProcessorList([transformers.TemperatureLogitsWarper(0.2)])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sample(proto, ids_d, logits_warper=warp, stopping_criteria=crit, **kw)
        torch.cuda.synchronize(); return time.perf_counter() - t0
    eager(0, 2); dropin(0, 2)
    te = sum(eager(q, n_new) for q in range(n_q))
    td = sum(dropin(q, n_new) for q in range(n_q))
    del m
    torch.cuda.empty_cache()
    what = ("installed transformers' LlamaForCausalLM + CLIPVisionModel composed like LlavaLlamaForCausalLM (tests/hf_llava.py), LLaVA-1.5-7B widths, "
            "32 + 24 layers, fp16, eager attention with output_attentions=True as llava_calibrate.py:175 passes it; B = 1, use_dd_unk")
    return ({"value": round(n_q * n_new / te, 2), "unit": "tokens/s", "kind": "reference stack (HF eager) + oracle restatement of the patched sample()",
             "what": what, "sample": f"{n_q} questions x {n_new} new tokens in {te:.1f}s"},
            {"value": round(n_q * n_new / td, 2), "unit": "tokens/s", "what": "this package's drop-in sample() over the same HF object: " + what,
             "sample": f"{n_q} questions x {n_new} new tokens in {td:.1f}s"})


def bench_llava_bench_eos(eng, dev, n_q=90, max_new=512, n_eos=250):
    """BASELINE config #3's call shape on this GPU's engine (llava_sampling.py:100-116: open-ended answers, use_dd + use_dd_unk, top-p 0.9;
    one image per question, so no shared image prefixes): answer lengths geometric (a random EOS set: ~0.8 % per step, capped at max_new),
    with and without row retirement + growing own-KV pools (VddLlavaEngine.retire).  tokens/s counts answer tokens only."""
    import numpy as np
    import torch
    ids, imgs = pope_prompts(n_q, per_img=1, seed=777)
    imgs = [im.to(dev).to(eng.dtype) for im in imgs]
    eos = sorted(set(np.random.default_rng(5).integers(3, eng.cfg.lm.vocab, size=n_eos).tolist()))
    kw = dict(images=imgs, use_dd=True, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=1.0, top_p=0.9, max_new_tokens=max_new,
              eos_token_id=eos, pad_token_id=0, seed=11, sync_every=8)
    eos_t = torch.tensor(eos, device=dev)
    out = {}
    for name, on in (("static", False), ("retire", True)):
        eng._kvs.clear(); eng._graphs.clear(); eng._kv = None
        torch.cuda.empty_cache()
        eng.retire = on
        torch.cuda.reset_peak_memory_stats(dev)                                   # (peak over warm-up + timed call: the pools of the two are the same)
        eng.generate(ids, **kw)                                                    # warm-up: tuner picks, graph capture paths
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter(); o = eng.generate(ids, **kw); torch.cuda.synchronize(dev); dt = time.perf_counter() - t0
        is_eos = (o.tokens[:, :, None] == eos_t).any(-1)
        lens = torch.where(is_eos.any(1), is_eos.float().argmax(1) + 1, torch.full((n_q,), o.tokens.shape[1], device=dev)).float()
        out[name] = {"tokens_per_s": round(float(lens.sum()) / dt, 1), "seconds": round(dt, 2), "decode_steps": int(o.tokens.shape[1]),
                     "mean_answer_tokens": round(float(lens.mean()), 1), "max_answer_tokens": int(lens.max()),
                     "hbm_peak_GB": round(torch.cuda.max_memory_allocated(dev) / 1e9, 1),
                     **({"retire_events": o.stats["retire_events"], "rows_at_end": o.stats["rows_at_end"]} if on else {})}
    eng.retire = True
    eng._kvs.clear(); eng._graphs.clear(); eng._kv = None
    torch.cuda.empty_cache()
    out["speedup"] = round(out["retire"]["tokens_per_s"] / out["static"]["tokens_per_s"], 2)
    # A LIST four times as long (llava_sampling.py:78 walks all of LLaVA-Bench): batch after batch through generate() with retirement, against
    # generate_list() with the same 90 questions in flight - waiting questions take the slots of finished ones
    ids4, imgs4 = pope_prompts(4 * n_q, per_img=1, seed=778)
    imgs4 = [im.to(dev).to(eng.dtype) for im in imgs4]

    def answer_tokens(tokens):
        hit = (tokens[:, :, None] == eos_t).any(-1)
        return float(torch.where(hit.any(1), hit.float().argmax(1) + 1, torch.full((tokens.shape[0],), tokens.shape[1], device=dev)).sum())

    def batches():
        return [eng.generate(ids4[b:b + n_q], **dict(kw, images=imgs4[b:b + n_q])) for b in range(0, 4 * n_q, n_q)]
    kwl = {k: v for k, v in kw.items() if k != "images"}
    eng.generate(ids4[:n_q], **dict(kw, images=imgs4[:n_q], max_new_tokens=160))      # warm-up: graph capture paths of this shape (the tuner's picks are already there)
    outs, dt_b = _timed(batches, dev)
    n_b = sum(answer_tokens(o.tokens) for o in outs)
    eng._kvs.clear(); eng._graphs.clear(); eng._kv = None
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(dev)
    eng.generate_list(ids4[:n_q + 40], imgs4[:n_q + 40], in_flight=n_q, **kwl)     # warm-up on a short list: captures, tuner picks of the admission waves
    ol, dt_l = _timed(lambda: eng.generate_list(ids4, imgs4, in_flight=n_q, **kwl), dev)
    out["list_of_360"] = {"batch_after_batch": {"tokens_per_s": round(n_b / dt_b, 1), "seconds": round(dt_b, 2), "answer_tokens": int(n_b)},
                          "generate_list": {"tokens_per_s": round(answer_tokens(ol.tokens) / dt_l, 1), "seconds": round(dt_l, 2), "answer_tokens": int(answer_tokens(ol.tokens)),
                                            "admissions": ol.stats["admissions"], "decode_steps": ol.stats["steps"], "mean_live_rows": ol.stats["mean_live_rows"],
                                            "hbm_peak_GB": round(torch.cuda.max_memory_allocated(dev) / 1e9, 1)},
                          "workload": f"{4 * n_q} questions of the same shape; both keep {n_q} questions = {3 * n_q} rows in flight (sampled runs: the two draw different "
                                      "random streams, their length samples differ slightly)"}
    out["list_of_360"]["speedup"] = round(out["list_of_360"]["generate_list"]["tokens_per_s"] / out["list_of_360"]["batch_after_batch"]["tokens_per_s"], 2)
    eng._kvs.clear(); eng._graphs.clear(); eng._kv = None
    torch.cuda.empty_cache()
    out["workload"] = (f"{n_q} questions x 3 branches (use_dd + use_dd_unk) = {3 * n_q} rows, one image each, top-p 0.9, T = 1, max_new_tokens {max_new}, "
                       f"{len(eos)} random EOS ids (sampled runs: the two legs draw different random streams, so their length samples differ slightly)")
    return out


def bench_cpu(eng):
    """SURVEY §8(d) CPU comparators, all fully measured on this host:
    (b) BASELINE config #1 end to end - 32 POPE-like questions, B=1, use_dd, top-k 1, 8 new tokens, toy LM with V = 32000 - through
        the oracle restatement of the reference loop (value);
    (a) the sampling tail with a stubbed forward at V = 32000 (use_dd_unk, T = 0.2), ms per step, comparable with BASELINE.md §2;
    and `value`: the headline's own model on the host - LLaVA-1.5-7B shapes, 1 question, 2 new tokens, all 32 layers measured when a
    2-layer estimate says that fits ~75 s (else the estimate, labelled)."""
    import copy
    import numpy as np
    import torch
    from oracle import vdd_oracle as O
    from toy_lm import BankModel, ToyVLM
    threads = torch.get_num_threads()
    # (b) config #1
    rng = np.random.default_rng(1234)
    toy = ToyVLM(vocab=32000, d=64, n_img=16, img_dim=12, seed=0)
    n_q, n_new, tot = 32, 8, 0.0
    for q in range(n_q):
        ids = torch.tensor([[1] + rng.integers(3, 31999, size=34).tolist() + [-200] + rng.integers(3, 31999, size=24).tolist()])
        img = torch.randn(1, 3, 2, 2, generator=torch.Generator().manual_seed(q // 6))
        t0 = time.perf_counter()
        O.reference_loop(toy, ids, warp=O.WarpConfig(top_k=1), max_length=ids.shape[1] + n_new, pad_token_id=None, eos_token_id=None,
                         pick=O.pick_multinomial, images=img, attention_mask=torch.ones_like(ids), use_cache=True, cd_alpha=1.0,
                         cd_beta=0.1, use_dd=True)
        tot += time.perf_counter() - t0
    cfg1 = n_q * n_new / tot
    # (a) sampling tail, stubbed forward: a model that replays pre-generated logit rows
    steps, V = 100, 32000
    g = torch.Generator().manual_seed(0)
    bank = [(torch.randn(1, V, generator=g) * 4).to(torch.bfloat16) for _ in range(2 * (steps + 2))]
    model = BankModel(bank)
    ids = torch.tensor([[1, 5, -200, 9]])
    t0 = time.perf_counter()
    O.reference_loop(model, ids, warp=O.WarpConfig(temperature=0.2), max_length=ids.shape[1] + steps, pad_token_id=None, eos_token_id=None,
                     pick=O.pick_multinomial, images=torch.zeros(1, 3, 2, 2), attention_mask=torch.ones_like(ids), use_cache=True,
                     cd_alpha=1.0, cd_beta=0.1, use_dd_unk=True)
    tail_ms = (time.perf_counter() - t0) / steps * 1e3
    # bounded 7B sample (labelled extrapolation)
    from llava_align_amd.engine import LlavaWeights
    layers, n7 = 2, 2
    cfg = copy.deepcopy(eng.cfg)
    full = cfg.lm.n_layers
    cfg.lm.n_layers = layers
    w = LlavaWeights(cfg, "cpu")
    for k, t in eng.w.t.items():
        if k.startswith("l") and k[1].isdigit() and int(k[1:k.index(".")]) >= layers:
            continue
        w.t[k] = t.cpu()
    pids, pimgs = pope_prompts(1, per_img=1, seed=99)
    t_small = reference_path(w, "cpu", pids[0], pimgs[0], n7)
    cfg0 = copy.deepcopy(cfg)
    cfg0.lm.n_layers = 0
    w0 = LlavaWeights(cfg0, "cpu")
    w0.t = {k: t for k, t in w.t.items() if not (k.startswith("l") and k[1].isdigit())}
    t_fixed = reference_path(w0, "cpu", pids[0], pimgs[0], n7)
    t_est = t_fixed + max(0.0, (t_small - t_fixed) / layers) * full
    # the SAME model as the headline, measured in full when the estimate says it fits the bench's CPU budget (it does on the GPU box's
    # host: ~30 s): all 32 decoder layers, ViT, projector, lm_head - one POPE-like question, use_dd_unk, 2 new tokens, B = 1
    measured = t_est <= 75.0
    if measured:
        wf = LlavaWeights(eng.cfg, "cpu")
        wf.t = {k: t.cpu() for k, t in eng.w.t.items()}
        t_full = reference_path(wf, "cpu", pids[0], pimgs[0], n7)
        del wf
    else:
        t_full = t_est
    how = (f"measured in full: all {full} decoder layers + ViT + projector + lm_head, {t_full:.1f}s (the {layers}-layer estimate said {t_est:.1f}s)" if measured else
           f"{layers} of {full} decoder layers timed ({t_small:.1f}s; depth-independent part {t_fixed:.1f}s), scaled to {full} layers -> {t_full:.1f}s (too slow to measure in full here)")
    return {"value": round(n7 / t_full, 4), "unit": "tokens/s", "cores": threads, "host_cpus": os.cpu_count(), "kind": "port",
            "comparable_with_headline": bool(measured), "extrapolated": not measured,
            "note": "value = the headline's own model and decoding mode on the host CPU (the oracle restatement of the reference's patched sample() over "
                    "this repo's plain-torch LLaVA, B = 1 as the reference decodes); the GPU figure beside it is `single_question` (same regime) - the "
                    "768-question headline batches what the reference cannot",
            "sample": f"LLaVA-1.5-7B shapes, 1 POPE-like question (35 sys + 576 image + ~24 text tokens), use_dd_unk, alpha 1, beta 0.1, T 0.2, {n7} new tokens, "
                      f"bf16 torch-CPU, {threads} threads: {how}",
            "config1": {"tokens_per_s": round(cfg1, 2),
                        "sample": f"BASELINE config #1, fully measured: {n_q} POPE-like questions (35 sys + image slot + 24 text tokens), B=1, use_dd, "
                                  f"top-k 1, {n_new} new tokens each, toy KV-cache LM (V=32000, d=64) on torch-CPU through the oracle restatement of "
                                  f"the reference loop: {tot:.1f}s (a DIFFERENT model from the headline: plumbing only)"},
            "sampling_tail": {"ms_per_step": round(tail_ms, 3), "V": 32000, "mode": "use_dd_unk, T=0.2, bf16 logits, forward stubbed (replayed rows)",
                              "steps": steps, "compare": "BASELINE.md §2: 1.4-2.4 ms/step on 8 vCPU"}}


def bench_fp16(a, dev, ids, host_imgs, n_new):
    """Secondary object of the bench line: the headline workload (same prompts, same kwargs) on an fp16 engine - every model kernel in
    its fp16 instantiation (csrc/vdd_elem.h), fp16 KV pools and logits - plus one question in flight, the reference's own regime."""
    import torch
    from llava_align_amd.engine import VddLlavaEngine
    eng = VddLlavaEngine(a.model, device=dev, seed=0, use_graph=True, lm_head_gain=4.0, dtype=torch.float16)
    on_dev = {}
    imgs = [on_dev.setdefault(id(im), im.to(dev).to(torch.float16)) for im in host_imgs]
    kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, max_new_tokens=n_new, seed=1, n_top=10)
    eng.generate(ids, **kw)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter(); eng.generate(ids, **kw); torch.cuda.synchronize(dev); dt = time.perf_counter() - t0
    kw2 = dict(kw, max_new_tokens=2)
    eng.generate(ids, **kw2)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter(); eng.generate(ids, **kw2); torch.cuda.synchronize(dev); t_pre = time.perf_counter() - t1
    ids1, imgs1 = pope_prompts(1, per_img=1, seed=99)
    kw1 = dict(images=imgs1, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, max_new_tokens=N_NEW, seed=3)
    eng.generate(ids1, **kw1); eng.generate(ids1, **kw1)
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter(); eng.generate(ids1, **kw1); torch.cuda.synchronize(dev); t_b1 = time.perf_counter() - t2
    Q = len(ids)
    return {"dtype": "fp16", "tokens_per_s_per_gpu": round(Q * n_new / dt, 1), "ms_per_step": round(dt * 1e3, 2),
            "decode_step_ms": round((dt - t_pre) / (n_new - 2) * 1e3, 3), "prefill_plus_first_token_s": round(t_pre, 4),
            "single_question_tokens_per_s": round(N_NEW / t_b1, 1), "hbm_peak_GB": round(torch.cuda.max_memory_allocated(dev) / 1e9, 1),
            "note": "the headline workload on VddLlavaEngine(dtype=torch.float16): the dtype the reference's drivers load (builder.py:40); one timed step "
                    "after one warm-up; same kernels compiled for fp16 storage (v_mfma_*_f16, v_dot2c_f32_f16, fp16 KV pools)"}


# ------------------------------------------------------------------ the other BASELINE configs, driver-timed (world == 1)
def _kv_bytes_per_token(lm, es=2):
    return 2 * lm.n_layers * lm.n_kv_heads * lm.head_dim * es


def step_roofline(eng, calls):
    """HBM roofline of the decode steps of the logged generate() calls that decoded more than one token (the main passes of a driver):
    SURVEY 8(d) bytes per step = LM weights once + K / V of the step's context + one logits row per decode row, over 8 TB/s, against
    the measured step (HIP events of VddLlavaEngine.call_log).  `frac` counts every SHARED prompt prefix once (the bytes that must
    cross the chip at least once per step); `frac_kv_per_row` is 8(d)'s literal sum over rows (it can exceed what the engine moves
    when rows share a prefix)."""
    lm, W = eng.cfg.lm, eng.w.lm_stream_bytes()
    kvb = _kv_bytes_per_token(lm)
    steps = ms = floor = floor_rows = rows = 0.0
    for st in calls:
        t = eng.call_timing(st)
        n = t["decode_steps"]
        if n < 1:
            continue
        R = st["decode_rows"]
        grow = R * (n + 1) / 2.0                                   # mean number of generated tokens in the context, summed over rows
        floor += n * (W + (st["ctx_tokens_distinct"] + grow) * kvb + R * lm.vocab * 2)
        floor_rows += n * (W + (st["ctx_tokens_rows"] + grow) * kvb + R * lm.vocab * 2)
        steps += n; ms += t["decode_ms"]; rows += R * n
    if not steps:
        return None
    step_ms = ms / steps
    return {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "decode_step_ms": round(step_ms, 3), "rows_per_step": round(rows / steps, 1),
            "floor_ms": round(floor / steps / (HBM_PEAK_GBS * 1e6), 3), "frac": round(floor / (HBM_PEAK_GBS * 1e6) / ms, 4),
            "frac_kv_per_row": round(floor_rows / (HBM_PEAK_GBS * 1e6) / ms, 4), "bytes_per_step": int(floor / steps),
            "lm_weight_bytes": W, "decode_steps_timed": int(steps),
            "note": "bytes = LM weights once + K/V of the context (shared prefixes once) + logits rows, SURVEY 8(d); measured with HIP events around the decode loop"}


def _timed(fn, dev):
    import torch
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(dev)
    return out, time.perf_counter() - t0


def _release(*engines):
    import gc
    import torch
    for e in engines:
        e._kvs.clear(); e._graphs.clear(); e._kv = None; e.call_log = None
    gc.collect()
    torch.cuda.empty_cache()


def _split_calls(eng, log):
    t = [eng.call_timing(st) for st in log]
    return {"prefill_s": round(sum(x["prefill_ms"] for x in t) / 1e3, 3), "decode_s": round(sum(x["decode_ms"] for x in t) / 1e3, 3), "generate_calls": len(log)}


def bench_config3(dev, n_q=90, n_new=256):
    """BASELINE config #3 (llava_sampling.py:96-109): LLaVA-1.5-13B shapes, LLaVA-Bench-like open generation - 90 questions, one image each,
    text 80 +- 30 tokens, use_dd + use_dd_unk (3 branches), top-p 0.9, T = 1, 256 new tokens - (a) the whole list on this GPU (270 rows)
    and (b) the questions ShardPlan gives rank 0 of 8 (ceil-chunks of whole images, MME/run_llava.py:32-40: 12 questions = 36 rows),
    i.e. the slowest rank of the 8-GPU run, whose time IS the 8-GPU job's time (no data-path collective)."""
    import torch as _t
    _t.cuda.reset_peak_memory_stats(dev)
    import numpy as np
    import torch
    from llava_align_amd.engine import VddLlavaEngine
    from llava_align_amd.shard import ShardPlan
    rng = np.random.default_rng(5)
    sys_tok = [1] + rng.integers(3, 32000, size=34).tolist()
    g = torch.Generator().manual_seed(3)
    ids, imgs = [], []
    for _ in range(n_q):
        n = int(np.clip(rng.normal(80, 30), 10, 170))
        ids.append(torch.tensor(sys_tok + [-200] + rng.integers(3, 32000, size=n).tolist()))
        imgs.append(torch.randn(3, 336, 336, generator=g).to(dev).to(torch.bfloat16))
    eng = VddLlavaEngine("llava-1.5-13b", device=dev, seed=0, use_graph=True)
    kw = dict(use_dd=True, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=1.0, top_p=0.9, max_new_tokens=n_new, seed=1)
    out = {"workload": f"LLaVA-1.5-13B shapes (40 layers, d 5120, synthetic weights), {n_q} questions x 3 branches (use_dd + use_dd_unk), one 336 px image each, "
                       f"text 80 +- 30 tokens, top-p 0.9, T = 1, {n_new} new tokens (no EOS), ViT + prefill + decode inside the timed call"}
    mine = list(ShardPlan([f"im{i}" for i in range(n_q)], 0, 8).mine)
    for name, sel in (("one_gpu", list(range(n_q))), ("rank0_of_8", mine)):
        a, b = [ids[i] for i in sel], [imgs[i] for i in sel]
        eng.generate(a, images=b, **kw)                                      # warm-up: tuner picks, graph capture
        eng.call_log = []
        o, dt = _timed(lambda: eng.generate(a, images=b, **kw), dev)
        out[name] = {"questions": len(sel), "rows_per_step": 3 * len(sel), "seconds": round(dt, 3), "tokens_per_s": round(len(sel) * n_new / dt, 1),
                     **_split_calls(eng, eng.call_log), "step_roofline": step_roofline(eng, eng.call_log)}
        eng.call_log = None
    t8 = out["rank0_of_8"]["seconds"]
    out["expected_8_gpus"] = {"tokens_per_s": round(n_q * n_new / t8, 1), "speedup_vs_one_gpu": round(out["one_gpu"]["seconds"] / t8, 2),
                              "note": f"every rank decodes its own {len(mine)}-question chunk (the last one {n_q - 7 * len(mine)}): job time = rank 0's time above + one result "
                                      "gather; strong scaling of a 90-question list is bounded by the 36-row step, not by the fabric"}
    out["hbm_peak_GB"] = round(torch.cuda.max_memory_allocated(dev) / 1e9, 1)
    _release(eng)
    del eng
    return out


_word = lambda t: ("yes", "no", "maybe")[t % 3]
_decode_words = lambda ids: " ".join(_word(t) for t in ids)


def bench_config4(dev, n_items=504):
    """BASELINE config #4 (MME/run_qwen.py:190-221): Qwen-VL-7B LM shape (32 layers, d 4096, qkv bias, V = 151,936), MME-like: 504 items =
    252 images x 2 questions, prompt = '<img>' + 256 resampler slots + ~40 text tokens as embeddings (the Qwen ViT + resampler are
    upstream of this path), use_dd_unk dual pass (the image-free branch re-runs the same inputs), 20 new tokens,
    min_new_tokens 1, pad = eos = eod, step-0 top-10, + the two text-only prior passes, the calibrate converter and the MME scorer."""
    import torch as _t
    _t.cuda.reset_peak_memory_stats(dev)
    import numpy as np
    import torch
    from llava_align_amd.engine import LlavaWeights, VddLlavaEngine, preset
    from llava_align_amd.mme_driver import MME_SUBSETS, qwen_mme_inputs, run_mme
    cfg = preset("qwen-vl-7b-lm")
    eng = VddLlavaEngine(cfg, weights=LlavaWeights.random(cfg, dev, seed=0, lm_head_gain=4.0), device=dev, use_graph=True)
    qs, gt = [], {}
    for i in range(n_items // 2):
        cat = MME_SUBSETS[i % 8]
        for k in range(2):
            text = f"Is item {i} {k} shown in the picture? Please answer yes or no."
            qs.append({"question_id": f"{cat}/{i:04d}.png", "image": f"{cat}/{i:04d}.png", "category": cat, "text": text})
            gt[(cat, f"{i:04d}.txt", text)] = ("Yes", "No")[(i + k) % 2]
    table = eng.w.t["embed"]
    g = torch.Generator(device=dev).manual_seed(2)
    feats, lead = {}, table[torch.tensor([151857, 151857], device=dev)]

    def embed_prompt(text, path):
        rng = np.random.default_rng(sum(map(ord, text)))                 # the same prompt embeds the same way in every pass
        e = table[torch.from_numpy(rng.integers(3, 151000, size=40 + (len(text) % 9))).to(dev)]
        if path is None:
            return e
        if path not in feats:
            feats[path] = (torch.randn(256, cfg.lm.d, device=dev, generator=g) * 0.02).to(torch.bfloat16)
        return torch.cat([lead, feats[path], e[2:]], 0), 258              # '<img>' + the 256 slots: shared by both questions about the image
    root = os.path.join(ROOT, "gpurun_out", "bench_mme") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else "/tmp/bench_mme"
    kw = dict(batch_questions=n_items, max_new_tokens=20, min_new_tokens=1, eos_token_id=151643, pad_token_id=151643, gt=gt, results_root=root,
              experiment="qwen", answers_path=os.path.join(root, "answers.jsonl"), use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=1.0, seed=1)
    build = qwen_mme_inputs(embed_prompt)
    run_mme(eng, qs, build, _decode_words, **kw)
    eng.call_log = []
    res, dt = _timed(lambda: run_mme(eng, qs, build, _decode_words, **kw), dev)
    out = {"workload": f"Qwen-VL-7B LM shape (V 151,936, qkv bias, synthetic weights), {n_items} MME-like items = {n_items // 2} images x 2, prompt = 258 shared image rows + ~40 text "
                       "rows as embeddings, use_dd_unk dual pass, 20 new tokens, min_new_tokens 1, + none / unk prior passes (1 token), converter + scorer (mme_driver.run_mme)",
           "items": len(qs), "seconds": round(dt, 3), "items_per_s": round(len(qs) / dt, 1),
           "main_pass_tokens_per_s": round(sum(len(a["text"].split()) for a in res["answers"]) / dt, 1), **_split_calls(eng, eng.call_log),
           "host_s": None, "step_roofline": step_roofline(eng, eng.call_log), "scored_subsets": sorted(k for k, v in res["scores"].items() if v is not None),
           "hbm_peak_GB": round(torch.cuda.max_memory_allocated(dev) / 1e9, 1)}
    out["host_s"] = round(dt - out["prefill_s"] - out["decode_s"], 3)
    # the reference runs the question file once per sampling setting (run_qwen.py:268-304: 51 settings); five of them in ONE pass (run_mme `sweep`:
    # the settings of a batch share its prefills) against five runs
    sw = [dict(tag=f"temp_{t}", temperature=t, top_p=None, top_k=None) for t in (0.2, 0.4, 0.6, 0.8, 1.0)]
    kws = {k: v for k, v in kw.items() if k not in ("temperature", "answers_path")}
    run_mme(eng, qs, build, _decode_words, sweep=sw, **kws)
    eng.call_log = []
    rs, dts = _timed(lambda: run_mme(eng, qs, build, _decode_words, sweep=sw, **kws), dev)
    out["sweep_of_5"] = {"seconds": round(dts, 3), "setting_items_per_s": round(5 * len(qs) / dts, 1), "vs_five_runs": round(5 * dt / dts, 2),
                         "generate_calls": len(eng.call_log), "prefills_reused": sum(bool(c.get("prefill_reused")) for c in eng.call_log),
                         "note": "five temperatures over the same items in one pass: per batch and pass type one prefill, five decodes"}
    _release(eng)
    del eng
    return out


def bench_config5(eng, dev, n_q=384):
    """BASELINE config #5 (blip_calibrate.py:78-98): InstructBLIP-Vicuna-7B shape - EVA-ViT-g (39 x 1408) + Q-Former (12 x 768) + the
    Vicuna-7B LM of `eng` - POPE-like: 384 questions (6 per 224 px image) in batches of 128, VCD branch from add_diffusion_noise(image, 500),
    alpha 0.5 (the driver never forwards cd_alpha), beta 0.1, top-p 1, top-k 50 (HF's default), max_length 20, noise(999) / zeros priors."""
    import torch as _t
    _t.cuda.reset_peak_memory_stats(dev)
    import torch
    from llava_align_amd.blip_driver import run_blip_pope
    from llava_align_amd.blip_frontend import BlipConfig, BlipWeights, InstructBlipFrontEnd
    front = InstructBlipFrontEnd(BlipWeights.random(BlipConfig(), dev, seed=1))
    images = {f"im{i}.jpg": torch.randn(3, 224, 224, generator=torch.Generator().manual_seed(i)).to(dev) for i in range(n_q // 6)}
    qs = [{"question_id": i, "image": f"im{i // 6}.jpg", "text": f"Is there a thing number {i} in the image?", "label": ("yes", "no")[i % 2]} for i in range(n_q)]
    tok_llm = lambda p: [1] + [(sum(map(ord, w)) * 31 + 7) % 31990 + 3 for w in p.split()]
    tok_qf = lambda p: [101] + [(sum(map(ord, w)) * 17) % 30000 + 200 for w in p.split()][:30] + [102]
    kw = dict(batch_questions=128, use_cd=True, noise_step=500, cd_beta=0.1, max_length=20, seed=1)
    run_blip_pope(eng, front, qs[:128], tok_llm, tok_qf, _decode_words, lambda n: images[n], **kw)
    eng.call_log = []
    res, dt = _timed(lambda: run_blip_pope(eng, front, qs, tok_llm, tok_qf, _decode_words, lambda n: images[n], **kw), dev)
    out = {"workload": f"InstructBLIP-Vicuna-7B shape (EVA-ViT-g 39 x 1408, Q-Former 12 x 768, Vicuna-7B; synthetic weights), {n_q} POPE-like questions (6 per 224 px image), "
                       "VCD branch = add_diffusion_noise(image, 500), alpha 0.5, beta 0.1, top-p 1, top-k 50, max_length 20, + noise(999) / zeros prior passes "
                       "(blip_driver.run_blip_pope: 4 EVA-ViT + Q-Former passes per batch)",
           "items": n_q, "seconds": round(dt, 3), "items_per_s": round(n_q / dt, 1), **_split_calls(eng, eng.call_log),
           "front_end_and_host_s": None, "step_roofline": step_roofline(eng, eng.call_log), "n_answers": len(res["answers"]),
           "hbm_peak_GB": round(torch.cuda.max_memory_allocated(dev) / 1e9, 1)}
    out["front_end_and_host_s"] = round(dt - out["prefill_s"] - out["decode_s"], 3)
    eng.call_log = None
    del front
    _release(eng)
    return out


def bench_config2_full(eng, dev, n_items=3000, batch=768):
    """BASELINE config #2 as the reference's driver runs it (llava_calibrate.py:130-219): 3,000 POPE-like items (500 images x 6) through
    pope_driver.run_pope - main pass (use_dd_unk, alpha 1, beta 0.1, T 0.2, max_new_tokens 64, answers stopped by EOS after 1 - 2 tokens),
    the none prior pass (the unk prior's label dict is read off the main pass's unk branch: the same ids), label dicts, the JSONL answers file and the plain +
    calibrated scorers.  The EOS ids are the second tokens
    the same seeded list emits in an untimed pass (which is also the warm-up)."""
    import torch as _t
    _t.cuda.reset_peak_memory_stats(dev)
    import torch
    from llava_align_amd.pope_driver import run_pope
    ids, imgs = pope_prompts(n_items // 6, seed=2024)
    on_dev = {}
    images = {f"im{i}.jpg": on_dev.setdefault(i, imgs[6 * i].to(dev).to(torch.bfloat16)) for i in range(n_items // 6)}
    by_text = {f"q{i}": ids[i].tolist() for i in range(n_items)}
    qs = [{"question_id": i, "image": f"im{i // 6}.jpg", "text": f"q{i}", "label": ("yes", "no")[i % 2]} for i in range(n_items)]
    enc = lambda text, with_image: by_text[text] if with_image else [t for t in by_text[text] if t != -200]
    kw = dict(batch_questions=batch, unk_token_id=0, pad_token_id=0, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, seed=1, stop_str=None)
    probe = run_pope(eng, qs, enc, lambda t: " ".join(map(str, t)), lambda n: images[n], max_new_tokens=2, eos_token_id=None, **kw)
    eos = sorted({int(a["text"].split()[1]) for a in probe["answers"]})
    path = os.path.join(ROOT, "gpurun_out", "bench_pope_answers.jsonl") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else "/tmp/bench_pope_answers.jsonl"
    # one untimed batch in the timed pass's own form (max_new_tokens 64 sizes the own-KV pools: allocating ~100 GB inside the timed pass cost 1 - 3 s,
    # differently from box to box)
    run_pope(eng, qs[:batch], enc, _decode_words, lambda n: images[n], max_new_tokens=64, eos_token_id=eos, sync_every=2, **kw)
    eng.call_log = []
    res, dt = _timed(lambda: run_pope(eng, qs, enc, _decode_words, lambda n: images[n], answers_path=path, max_new_tokens=64, eos_token_id=eos,
                                      sync_every=2, **kw), dev)
    n_tok = sum(len(a["text"].split()) for a in res["answers"])
    sc = res["scores"]
    out = {"workload": f"LLaVA-1.5-7B shapes, {n_items} POPE-like items = {n_items // 6} images x 6 in batches of {batch} through pope_driver.run_pope: main pass (use_dd_unk, alpha 1, "
                       f"beta 0.1, T 0.2, max_new_tokens 64, {len(eos)} EOS ids -> answers of 1 - 2 tokens) + none prior pass (unk prior = the main pass's unk branch) + label dicts + JSONL + scorers",
           "items": n_items, "seconds": round(dt, 3), "items_per_s": round(n_items / dt, 1), "answer_tokens_per_s": round(n_tok / dt, 1),
           "mean_answer_tokens": round(n_tok / n_items, 2), **_split_calls(eng, eng.call_log), "host_s": None,
           "answers_file_bytes": os.path.getsize(path), "scorers_ran": sorted(k for k, v in sc.items() if v is not None),
           "nan_rows": {k: v.get("nan_rows") for k, v in sc.items() if isinstance(v, dict) and "nan_rows" in v}}
    out["host_s"] = round(dt - out["prefill_s"] - out["decode_s"], 3)
    eng.call_log = None
    _release(eng)
    return out


def bench_batch_invariant(eng, dev, ids, kw, baseline_s):
    """The headline step under ops.GEMM_BATCH_INVARIANT (data-parallel GEMM schedule only, fixed dispatch thresholds: a row's logits no
    longer depend on who else is in the batch - what the sharded drivers select so that 1-GPU and N-GPU cd_greedy runs agree)."""
    from llava_align_amd import ops
    old = ops.GEMM_BATCH_INVARIANT
    ops.GEMM_BATCH_INVARIANT = True
    try:
        _release(eng)
        eng.generate(ids, **kw)
        _, dt = _timed(lambda: eng.generate(ids, **kw), dev)
    finally:
        ops.GEMM_BATCH_INVARIANT = old
        _release(eng)
    Q, n_new = len(ids), kw["max_new_tokens"]
    return {"tokens_per_s_per_gpu": round(Q * n_new / dt, 1), "seconds_per_step": round(dt, 3), "cost_vs_tuned_schedules": round(dt / baseline_s, 4),
            "note": "same workload as the headline, ops.GEMM_BATCH_INVARIANT = True; one timed step after one warm-up"}


# ------------------------------------------------------------------ strong scaling: one question list split over the ranks
def run_strong(a, eng, dev, rank, world):
    """`--strong N`: what the eval drivers do on a node (MME/run_llava.py:32-40): ONE seeded POPE-like list of N questions,
    rank k takes get_chunk(N, world, k, group=6) (contiguous ceil-chunks of whole image groups), decodes it in batches of
    --questions with EOS on (answers of 1-2 tokens: the EOS ids are the second tokens a 2-token probe of the shard emits), and
    the per-question payload is gathered ONCE; rank 0 checks that the gathered question ids are 0..N-1 exactly once.
    value = answer tokens of the whole list / max-over-ranks wall time, per timed step = one pass over the whole list."""
    import torch
    import torch.distributed as dist
    from llava_align_amd.shard import gather_results, get_chunk
    N = a.strong - a.strong % 6
    ids_all, imgs_all = pope_prompts(N // 6, seed=4321, vocab=eng.cfg.lm.vocab, image=eng.cfg.vision.image)      # same list on every rank
    mine = get_chunk(N, world, rank, group=6)
    on_dev = {}
    ids = [ids_all[i] for i in mine]
    imgs = [on_dev.setdefault(id(imgs_all[i]), imgs_all[i].to(dev).to(torch.bfloat16)) for i in mine]
    n_new = N_NEW if not a.model.startswith("tiny") else 16
    kw = dict(use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, seed=1 + rank, n_top=10)
    batches = [range(b, min(b + a.questions, len(ids))) for b in range(0, len(ids), a.questions)]
    probe = [eng.generate([ids[i] for i in b], images=[imgs[i] for i in b], max_new_tokens=2, **kw).tokens[:, 1] for b in batches]
    eos = sorted(set(torch.cat(probe).tolist())) if probe else [0]

    def one_pass():
        outs = [eng.generate([ids[i] for i in b], images=[imgs[i] for i in b], max_new_tokens=n_new, eos_token_id=eos, pad_token_id=0,
                             sync_every=2, **kw) for b in batches]
        T = max([o.tokens.shape[1] for o in outs] + [1])
        pad = lambda t: torch.nn.functional.pad(t, (0, T - t.shape[1]))
        toks = torch.cat([pad(o.tokens) for o in outs]) if outs else torch.zeros(0, T, dtype=torch.long, device=dev)
        tt = torch.cat([o.top_tok for o in outs]) if outs else torch.zeros(0, 10, dtype=torch.long, device=dev)
        tp = torch.cat([o.top_prob for o in outs]) if outs else torch.zeros(0, 10, device=dev)
        return toks, tt, tp
    for _ in range(a.warmup):
        one_pass()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        toks, tt, tp = one_pass()
    eos_t = torch.tensor(eos, device=dev)
    hit = (toks[:, :, None] == eos_t[None, None, :]).any(-1)
    n_tok = torch.where(hit.any(1), hit.float().argmax(1) + 1, torch.full((toks.shape[0],), toks.shape[1], device=dev))
    cap = max(len(get_chunk(N, world, k, group=6)) for k in range(world))             # known on every rank: no size exchange
    res = gather_results(torch.tensor(list(mine), dtype=torch.long, device=dev), toks, n_tok, tt, tp, N, capacity=cap, width=n_new)   # ragged T across ranks
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    if rank == 0:
        assert res["count"].tolist() == [1] * N, "the shards do not partition the question list"
        total_tokens = int(res["n_tokens"].sum().item())
        print(json.dumps({"metric": "decode tokens/sec (VDD dual-pass) LLaVA-1.5-7B POPE", "value": round(total_tokens * a.steps / dt, 1), "unit": "tokens/s",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True,
                          "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                          "config": {"workload": f"ONE list of {N} POPE-like questions ({N // 6} images x 6) split over the ranks by get_chunk(group=6), "
                                                 f"use_dd_unk, cd_alpha=1, cd_beta=0.1, T=0.2, EOS after 1-2 tokens, batches of {a.questions}",
                                     "questions_total": N, "questions_rank0": len(ids), "parallelism": f"dp{world}"},
                          "questions_per_s": round(N * a.steps / dt, 1), "mean_answer_tokens": round(total_tokens / N, 2),
                          "gathered_question_ids_cover_the_list_once": True, "roofline": None, "cpu_baseline": None}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _leg(line, name, fn, *args):
    """A secondary leg must never cost the bench its line: an exception is recorded under the leg's name (and on stderr) and the
    next leg runs on a cleaned-up device."""
    import gc
    import traceback
    import torch
    try:
        line[name] = fn(*args)
    except Exception as e:                                   # noqa: BLE001 (reported, not swallowed: the line says which leg failed and why)
        traceback.print_exc()
        line[name] = {"error": f"{type(e).__name__}: {e}"[:500]}
        gc.collect()
        torch.cuda.empty_cache()


# ------------------------------------------------------------------ main
def main():
    a = parse_args()
    maybe_relaunch(a)
    import torch
    rank, local, world = dist_env()
    if world != a.gpus and "RANK" in os.environ and rank == 0:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; using the launcher's world size", file=sys.stderr)
    if os.environ.get("VDD_FORCE_DEVICE") is not None:      # dry-run aid: several ranks on one GPU (with VDD_DIST_BACKEND=gloo)
        local = int(os.environ["VDD_FORCE_DEVICE"])
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    # VDD_FORCE_DIST=1: initialise the process group (and run barrier / all_reduce / the result all_gather through it) even at world
    # size 1 - the only way to execute the RCCL code path on a single-GPU box (tests/test_bench_gpu.py)
    use_dist = world > 1 or os.environ.get("VDD_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        backend = os.environ.get("VDD_DIST_BACKEND", "nccl")            # "nccl" is RCCL on ROCm
        if "RANK" not in os.environ:
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"))
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
        assert dist.get_world_size() == world

    import llava_align_amd  # noqa: F401  (raises if the HIP library is missing)
    from llava_align_amd.engine import VddLlavaEngine
    from llava_align_amd.shard import gather_results

    tiny = a.model.startswith("tiny")
    roof = roof_extra = None
    if rank == 0 and not tiny and not a.strong:
        roof = kernel_point(dev, 4096, 32000)
        roof_extra = [kernel_point(dev, 4096, 32000, beta=1e-6, iters=50), kernel_point(dev, 1024, 151936, iters=50),
                      kernel_point(dev, 1024, 151936, scores=False, iters=50),
                      # rows that stay in global memory run four workgroups per CU since round 5 (1,024 resident rows: one round at B = 1,024;
                      # before: 768 resident, 105 us -> 94 us); 3,072 rows = three full rounds (profiles/r05_kernel_points_qwen.jsonl)
                      kernel_point(dev, 3072, 151936, scores=False, iters=30),
                      # the launches the ENGINE makes on the headline workload: one row per question (768) - and 1,536 for a batch twice
                      # the size; 768 rows are exactly one resident round (3 workgroups x 256 CUs), so launch overhead and the ramp of a
                      # single round weigh more than at B = 4096
                      kernel_point(dev, 768, 32000, iters=100), kernel_point(dev, 1536, 32000, iters=100)]
    eng = VddLlavaEngine(a.model, device=dev, seed=0, use_graph=True, lm_head_gain=4.0)
    if a.strong:
        return run_strong(a, eng, dev, rank, world)
    n_img = max(1, a.questions // 6)
    # every rank: its own shard of the question list (weak scaling)
    ids, imgs = pope_prompts(n_img, seed=1234 + rank, vocab=eng.cfg.lm.vocab, image=eng.cfg.vision.image)
    # inputs resident in HBM before the timed region (the reference's drivers also hand generate() device tensors:
    # `image_tensor.unsqueeze(0).half().cuda()`, llava_calibrate.py:163); one device tensor per DISTINCT image, shared by its
    # 6 questions.  The host-image (PCIe-inclusive) rate is reported separately as `pcie_inclusive`.
    host_imgs = imgs
    on_dev = {}
    imgs = [on_dev.setdefault(id(im), im.to(dev).to(torch.bfloat16)) for im in host_imgs]
    Q = len(ids)
    n_new = N_NEW if not tiny else 16
    kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, max_new_tokens=n_new, seed=1 + rank, n_top=10)

    def step():
        return eng.generate(ids, **kw)

    for _ in range(a.warmup):
        out = step()
    torch.cuda.synchronize(dev)
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    res = gather_results(torch.arange(rank * Q, (rank + 1) * Q, device=dev), out.tokens, torch.full((Q,), out.tokens.shape[1], device=dev),
                         out.top_tok, out.top_prob, world * Q, capacity=Q, width=n_new)   # the one result gather (one collective)
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    assert res["tokens"].shape == (world * Q, n_new) and res["top_prob"].shape == (world * Q, 10)

    if rank == 0:
        # decode-only rate: same batch, 2 new tokens (prefill + 1 decode step) subtracted
        kw2 = dict(kw, max_new_tokens=2)
        eng.generate(ids, **kw2); eng.generate(ids, **kw2)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter(); o2 = eng.generate(ids, **kw2); torch.cuda.synchronize(dev); t_pre = time.perf_counter() - t1
        ms_step = dt / a.steps * 1e3
        ms_decode = (dt / a.steps - t_pre) / (n_new - 2) * 1e3
        wbytes = eng.w.lm_stream_bytes()
        if roof is not None:
            roof["traffic_pmc"] = pmc_traffic()
        line = {"metric": "decode tokens/sec (VDD dual-pass) LLaVA-1.5-7B POPE", "value": round(world * Q * n_new * a.steps / dt, 1),
                "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 2),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": f"LLaVA-1.5-7B shapes (synthetic N(0,0.02) weights, lm_head x4), POPE-like: {Q} questions/GPU = {n_img} images x 6, "
                                       f"prompt 35 sys + 576 image + 19..28 text tokens, use_dd_unk, cd_alpha=1, cd_beta=0.1, T=0.2, "
                                       f"{n_new} new tokens (no EOS), ViT + both-branch prefill + decode + step-0 top-10 all inside the timed step",
                           "questions_per_gpu": Q, "max_new_tokens": n_new, "parallelism": f"dp{world}"},
                "tokens_per_s_per_gpu": round(Q * n_new * a.steps / dt, 1),
                "decode_only_tokens_per_s_per_gpu": round(Q / (ms_decode * 1e-3), 1),
                "prefill_plus_first_token_s": round(t_pre, 4),
                "decode_step": {"ms": round(ms_decode, 3), "rows": 2 * Q, "lm_weight_bytes": wbytes,
                                "weight_stream_GBs": round(wbytes / (ms_decode * 1e-3) / 1e9, 1),
                                "note": "per step the LM weights are streamed once for all rows; KV reads come on top"},
                "hbm_peak_GB": round(torch.cuda.max_memory_allocated(dev) / 1e9, 1),
                "prefill_tokens": out.stats["prefill_tokens"], "unshared_prefill_tokens": out.stats["unshared_prefill_tokens"],
                "gemm": "hand-written MFMA kernels for every projection (csrc/vdd_gemm.hip; no hipBLASLt on the path)",
                "roofline": roof, "roofline_extra": roof_extra}
        # POPE proper: answers are 1-2 tokens, then EOS.  The EOS ids are the second tokens this very (seeded) batch emits, so every
        # question stops through the kernel's EOS / pad / unfinished logic after 1 or 2 tokens; max_new_tokens stays 64.
        eos = sorted(set(o2.tokens[:, 1].tolist()))
        kwe = dict(kw, eos_token_id=eos, pad_token_id=0, sync_every=2)
        eng.generate(ids, **kwe)
        torch.cuda.synchronize(dev)
        t_es = []
        for _ in range(3):                               # median of three: a single run of this short leg carries a GC pause or not (1.41 vs 1.56 s seen)
            t5 = time.perf_counter(); oe = eng.generate(ids, **kwe); torch.cuda.synchronize(dev); t_es.append(time.perf_counter() - t5)
        t_e = sorted(t_es)[1]
        eos_t = torch.tensor(eos, device=dev)
        ans_len = ((oe.tokens[:, :, None] == eos_t[None, None, :]).any(-1).float().argmax(1) + 1).float()
        line["pope_eos"] = {"questions_per_s_per_gpu": round(Q / t_e, 1), "tokens_per_s_per_gpu": round(float(ans_len.sum()) / t_e, 1),
                            "mean_answer_tokens": round(float(ans_len.mean()), 2), "decode_steps_run": int(oe.tokens.shape[1]),
                            "seconds_per_batch": round(t_e, 4),
                            "note": f"{len(eos)} EOS ids = the second tokens of this seeded batch: every question emits EOS after 1-2 tokens "
                                    "(POPE answers); the run ends when the device-side `unfinished` vector is all zero (checked every 2 steps)"}
        if world == 1 and not tiny:
            # ... and with POPE's question statistics: the benchmark asks "Is there a <object> in the image?" for ~80 object names about 500 images,
            # so the image-free rows of a batch repeat (engine._plan shares everything but the last position of identical rows).  Same batch, the
            # 768 question texts drawn from 80 distinct ones.  NOT the headline workload (SURVEY 8d draws every text at random): reported beside it.
            import numpy as _np
            pick = _np.random.default_rng(5).integers(0, 80, size=Q).tolist()
            texts = [r[r.tolist().index(-200) + 1:] for r in ids[:80]]
            ids_rep = [torch.cat([r[: r.tolist().index(-200) + 1], texts[j]]) for r, j in zip(ids, pick)]
            o2r = eng.generate(ids_rep, **dict(kw, max_new_tokens=2))
            kwr = dict(kw, eos_token_id=sorted(set(o2r.tokens[:, 1].tolist())), pad_token_id=0, sync_every=2)
            eng.generate(ids_rep, **kwr)
            torch.cuda.synchronize(dev)
            t_rs = []
            for _ in range(3):
                t6 = time.perf_counter(); orp = eng.generate(ids_rep, **kwr); torch.cuda.synchronize(dev); t_rs.append(time.perf_counter() - t6)
            line["pope_eos"]["repeated_questions"] = {"questions_per_s_per_gpu": round(Q / sorted(t_rs)[1], 1), "distinct_question_texts": 80,
                                                      "prefill_tokens": orp.stats["prefill_tokens"], "seconds_per_batch": round(sorted(t_rs)[1], 4),
                                                      "note": "the pope_eos batch with its 768 question texts drawn from 80 distinct ones (POPE's own statistics): "
                                                              "identical <unk>-branch rows share their prompt"}
        if world == 1:
            # the same step with the images handed over as host fp32 tensors (pageable): upload + cast inside the timed call
            kw_h = dict(kw, images=host_imgs)
            eng.generate(ids, **kw_h)
            torch.cuda.synchronize(dev)
            t3 = time.perf_counter(); eng.generate(ids, **kw_h); torch.cuda.synchronize(dev); t_h = time.perf_counter() - t3
            line["pcie_inclusive"] = {"value": round(Q * n_new / t_h, 1), "unit": "tokens/s",
                                      "note": f"images passed as host fp32 tensors ({n_img} x 1.35 MB pageable): upload and cast inside generate()"}
        if world == 1 and not a.no_baselines and not tiny:
            # single question in flight (the reference's own B=1 regime): latency-mode tokens/s of the engine
            ids1, imgs1 = pope_prompts(1, per_img=1, seed=99)
            kw1 = dict(images=imgs1, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, max_new_tokens=N_NEW, seed=3)
            eng.generate(ids1, **kw1); eng.generate(ids1, **kw1)
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter(); eng.generate(ids1, **kw1); torch.cuda.synchronize(dev); t_b1 = time.perf_counter() - t2
            line["single_question"] = {"tokens_per_s": round(N_NEW / t_b1, 1), "ms_per_token": round(t_b1 / N_NEW * 1e3, 2),
                                       "note": "B=1 (2 rows: main + <unk> branch), prefill + 64 tokens, HIP-graph decode"}
            # the >= 6x comparator (north_star): HF's own eager stack in the reference's dtype; the repo's plain-torch port stays beside it
            _leg(line, "_hf", bench_hf_gpu, dev)
            hf = line.pop("_hf")
            line["eager_gpu"], line["dropin_gpu"] = hf if isinstance(hf, tuple) else (hf, hf)
            _leg(line, "eager_gpu_port", bench_eager_gpu, eng, dev)
            _leg(line, "dropin_gpu_port", bench_dropin_gpu, eng, dev)
            if "value" in line["eager_gpu"]:
                line["speedup_vs_eager_gpu"] = round(line["value"] / line["eager_gpu"]["value"], 1)
                line["speedup_vs_eager_gpu_single_question"] = round(line["single_question"]["tokens_per_s"] / line["eager_gpu"]["value"], 1)
            _leg(line, "cpu_baseline", bench_cpu, eng)
            _leg(line, "llava_bench_eos", bench_llava_bench_eos, eng, dev)
            # every BASELINE config on the driver-timed line; the 7B engine serves #2-full, #5 and the batch-invariant step
            _leg(line, "batch_invariant", bench_batch_invariant, eng, dev, ids, kw, dt / a.steps)
            _leg(line, "config2_full", bench_config2_full, eng, dev)
            _leg(line, "config5", bench_config5, eng, dev)
            # the same workload in the reference's own dtype (fp16: builder.py:40; config #2 - the headline - says bf16): the bf16 engine
            # and its KV pools go first (two engines do not fit 288 GB at 768 questions)
            del eng, out, oe, o2
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            _leg(line, "fp16", bench_fp16, a, dev, ids, host_imgs, n_new)
            gc.collect()
            torch.cuda.empty_cache()
            _leg(line, "config3", bench_config3, dev)
            _leg(line, "config4", bench_config4, dev)
        else:
            line["cpu_baseline"] = None
        line["collective_backend"] = (os.environ.get("VDD_DIST_BACKEND", "nccl") if use_dist else None)
        if os.path.isdir(os.path.join(ROOT, "gpurun_out")):        # the tuner's picks of this run (source of llava-align_amd/gemm_choices_mi355x.json)
            from llava_align_amd import ops
            with open(os.path.join(ROOT, "gpurun_out", "gemm_choices_bench.json"), "w") as f:
                json.dump({"device": torch.cuda.get_device_name(dev), "choices": ops.gemm_choices_export()}, f, indent=0, sort_keys=True)
        print(json.dumps(line))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
