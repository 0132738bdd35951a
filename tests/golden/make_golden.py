"""Generate the golden fixtures by running the UNMODIFIED reference functions
(read-only import from /root/reference, see tests/ref_shim.py).  Build container only:

    python tests/golden/make_golden.py

Writes tests/golden/{kernel_vectors.npz,kernel_vectors.json,loop_traces.json,
eos_pad.json,noise.npz,calibration.json}.  Only inputs/outputs are stored — no
reference source.  Versions used are recorded in kernel_vectors.json.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))            # tests/
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))  # repo root
from golden.gen_inputs import DTYPES, logit_rows, to_bits  # noqa: E402
from ref_shim import load_reference, run_reference  # noqa: E402
from toy_lm import BankModel, ToyVLM, pope_like_ids  # noqa: E402

ARGMAX_MN = lambda probs, num_samples=1, **kw: probs.argmax(-1, keepdim=True)  # noqa: E731
MODE_KW = {1: {}, 2: {"use_dd_unk": True}, 3: {"use_dd": True, "use_dd_unk": True}}


def kernel_cases():
    cases = []

    def add(**kw):
        kw.setdefault("kind", "normal"); kw.setdefault("steps", 1); kw.setdefault("B", 3)
        kw.setdefault("warp", {}); kw.setdefault("alpha", 1.0); kw.setdefault("beta", 0.1)
        kw["seed"] = 1000 + len(cases)
        cases.append(kw)

    warps = [{}, {"temperature": 0.05}, {"temperature": 0.2}, {"top_k": 1}, {"top_k": 2}, {"top_k": 50},
             {"top_p": 0.05}, {"top_p": 0.6}, {"top_p": 0.9}, {"top_p": 0.0},
             {"temperature": 0.7, "top_k": 50, "top_p": 0.9}, {"temperature": 0.2, "top_k": 1}]
    # small-V full grid (V=97 and an odd, non-multiple-of-8 V=1003)
    for V in (97, 1003):
        for dt in ("fp32", "fp16", "bf16"):
            for n_in in (1, 2, 3):
                for wi, w in enumerate(warps):
                    for (a, b) in ((1.0, 0.1), (0.5, 0.2)):
                        if n_in == 1 and (a, b) != (1.0, 0.1):
                            continue
                        if V == 1003 and (wi % 3 != (0 if n_in == 2 else 1) or (a, b) != (1.0, 0.1)):
                            continue
                        add(V=V, dtype=dt, n_in=n_in, warp=w, alpha=a, beta=b, steps=2)
            for b in (1.0, 1e-6, 0.5):
                add(V=V, dtype=dt, n_in=2, beta=b, warp={"temperature": 0.2})
            for kind in ("flat", "vc_equal", "big", "one_left", "max_tie"):
                add(V=V, dtype=dt, n_in=2, kind=kind, warp={"temperature": 0.2})
                add(V=V, dtype=dt, n_in=3, kind=kind, warp={"top_p": 0.9}, alpha=0.5)
                add(V=V, dtype=dt, n_in=2, kind=kind, warp={"top_k": 2})
            add(V=V, dtype=dt, n_in=2, alpha=0.3, beta=0.37, warp={"temperature": 0.33})  # inexact scalars
    # real vocab sizes (inputs regenerated from the seed; outputs stored sparsely)
    for dt in ("fp16", "bf16"):
        add(V=32000, dtype=dt, n_in=2, B=2, warp={"temperature": 0.2})                 # POPE config
        add(V=32000, dtype=dt, n_in=3, B=2, warp={"top_p": 0.9})                       # LLaVA-Bench config
        add(V=32000, dtype=dt, n_in=2, B=2, warp={"top_k": 1})
        add(V=32000, dtype=dt, n_in=2, B=2, kind="flat", beta=0.2, warp={"temperature": 0.2, "top_k": 50})
        add(V=32000, dtype=dt, n_in=1, B=2, warp={"temperature": 0.7, "top_k": 50})    # plain path
        add(V=151936, dtype=dt, n_in=2, B=1, alpha=0.5, warp={"temperature": 0.2})     # Qwen vocab
    add(V=32000, dtype="fp32", n_in=2, B=2, warp={"temperature": 0.2})
    add(V=151936, dtype="bf16", n_in=3, B=1, kind="vc_equal", warp={"top_p": 0.6})
    return cases


def sparse_pack(t: torch.Tensor):
    """finite entries as (flat index, bit pattern); everything else must be -inf."""
    flat = t.reshape(-1)
    fin = torch.isfinite(flat)
    idx = torch.nonzero(fin).reshape(-1)
    return idx.numpy().astype(np.int64), to_bits(flat[idx]), int((flat == -float("inf")).sum()), \
        int(torch.isnan(flat).sum()), int((flat == float("inf")).sum())


class _GpuScalarEmulation:
    """Runs the REAL reference sample() on the CPU with torch-GPU's scalar arithmetic at the two places where the backends differ
    (found by probing torch-ROCm eager, tests/test_kernel_gpu.py::test_torch_gpu_eager_agrees_within_reference_tolerance):
      * vcd_sample.py:191 `torch.log(torch.tensor(cd_beta)) + max`: on a GPU the 0-dim CPU tensor is an fp32 scalar operand,
        fl_dtype(float(max) + fl32(log beta)); on the CPU it is demoted to the model dtype first.  Emulated by a Tensor subclass
        returned from torch.log for 0-dim inputs whose `+` does the fp32 add;
      * HF TemperatureLogitsWarper `scores / T`: on a GPU a multiplication by fl32(1 / T).  Emulated by a warper class that
        replaces it in the warper list.
    Everything else (the loop, contrast arithmetic, top-k / top-p warpers, softmax) is the unmodified reference / HF code."""

    class F32Scalar(torch.Tensor):
        @classmethod
        def __torch_function__(cls, func, types, args=(), kwargs=None):
            kwargs = kwargs or {}
            if func in (torch.Tensor.__add__, torch.Tensor.add, torch.add, torch.Tensor.__radd__):
                a, b = args[0], args[1]
                s_, t = (a, b) if isinstance(a, cls) else (b, a)
                with torch._C.DisableTorchFunctionSubclass():
                    s32 = s_.as_subclass(torch.Tensor).float()
                    return (t.float() + s32).to(t.dtype)
            with torch._C.DisableTorchFunctionSubclass():
                return func(*[x.as_subclass(torch.Tensor) if isinstance(x, cls) else x for x in args], **kwargs)

    class RecipTemperature:
        def __init__(self, t):
            self.inv = torch.tensor(1.0, dtype=torch.float32) / torch.tensor(float(t), dtype=torch.float32)

        def __call__(self, input_ids, scores):
            return (scores.float() * self.inv).to(scores.dtype)

    def __enter__(self):
        import ref_shim
        self._log, self._hw = torch.log, ref_shim.hf_warpers
        real_log, F32 = torch.log, self.F32Scalar

        def log(x, *a, **k):
            y = real_log(x, *a, **k)
            return y.as_subclass(F32) if (torch.is_tensor(x) and x.dim() == 0) else y
        torch.log = log
        real_hw, Recip = ref_shim.hf_warpers, self.RecipTemperature

        def hf_warpers(temperature=None, **kw):
            lst = real_hw(temperature=None, **kw)
            if temperature is not None and temperature != 1.0:
                lst.insert(0, Recip(temperature))
            return lst
        ref_shim.hf_warpers = hf_warpers
        return self

    def __exit__(self, *a):
        import ref_shim
        torch.log, ref_shim.hf_warpers = self._log, self._hw


def gen_kernel_vectors_gpu_scalar():
    """Second golden set: the same cases (V <= 32000; fp16 / bf16 - for fp32 the two backends agree) under torch-GPU scalar arithmetic."""
    with _GpuScalarEmulation():
        gen_kernel_vectors(out_name="kernel_vectors_gpu_scalar", keep=lambda c: c["dtype"] != "fp32" and c["V"] <= 32000)


def gen_kernel_vectors(out_name="kernel_vectors", keep=lambda c: True):
    arrays, manifest = {}, []
    for ci, case in enumerate(kernel_cases()):
        if not keep(case):
            continue
        dt = DTYPES[case["dtype"]]
        rows = logit_rows(case["seed"], case["B"], case["V"], dt, case["n_in"], case["kind"], case["steps"])
        bank = [r for step in rows for r in step]
        model = BankModel(bank)
        ids = torch.ones(case["B"], 4, dtype=torch.long)
        ids[:, 2] = -200
        kw = dict(attention_mask=torch.ones_like(ids), cd_alpha=case["alpha"], cd_beta=case["beta"],
                  **MODE_KW[case["n_in"]])
        out = run_reference(model, ids, max_length=4 + case["steps"], warp=case["warp"], multinomial=ARGMAX_MN, **kw)
        entry = dict(case)
        entry["id"] = ci
        entry["tokens"] = out.sequences[:, 4:].tolist()
        dense = case["V"] <= 1003
        entry["dense"] = dense
        # inputs are never stored: tests regenerate them with gen_inputs.logit_rows(seed, ...)
        for s, sc in enumerate(out.scores):
            assert sc.dtype == dt
            if dense:
                arrays[f"c{ci}_s{s}_scores"] = to_bits(sc)
            else:
                idx, bits, n_neg, n_nan, n_pos = sparse_pack(sc)
                if len(idx) > 4096:   # dense-finite rows: keep a strided sample
                    keep = np.arange(0, len(idx), 61)
                    entry.setdefault("strided", {})[str(s)] = 61
                    idx, bits = idx[keep], bits[keep]
                arrays[f"c{ci}_s{s}_idx"] = idx
                arrays[f"c{ci}_s{s}_val"] = bits
                entry.setdefault("counts", {})[str(s)] = [n_neg, n_nan, n_pos]
            entry.setdefault("sha256", {})[str(s)] = hashlib.sha256(to_bits(sc).tobytes()).hexdigest()
        manifest.append(entry)
    # error behaviour: beta > 1 masks every token -> softmax NaN -> multinomial raises (vcd_sample.py:191-202)
    rows = logit_rows(77, 1, 97, torch.float16, 2)[0]
    ids = torch.ones(1, 4, dtype=torch.long)
    try:
        run_reference(BankModel(rows), ids, max_length=5, warp={}, attention_mask=torch.ones_like(ids),
                      use_dd_unk=True, cd_alpha=1.0, cd_beta=2.0)
        raised = None
    except RuntimeError as e:
        raised = type(e).__name__
    import transformers
    meta = {"torch": torch.__version__, "transformers": transformers.__version__, "numpy": np.__version__,
            "all_masked_row_raises": raised, "cases": manifest}
    np.savez_compressed(os.path.join(HERE, out_name + ".npz"), **arrays)
    with open(os.path.join(HERE, out_name + ".json"), "w") as f:
        json.dump(meta, f, indent=0)
    print(out_name + ":", len(manifest), "cases")


def loop_kwargs(mode, ids, img, img_cd):
    kw = dict(images=img, attention_mask=torch.ones_like(ids), use_cache=True, cd_alpha=1.0, cd_beta=0.1)
    kw.update({"plain": {}, "cd": {"images_cd": img_cd}, "dd": {"use_dd": True}, "dd_unk": {"use_dd_unk": True},
               "both": {"use_dd": True, "use_dd_unk": True}}[mode])
    return kw


def gen_loop_traces():
    """32 POPE-like prompts, toy LM, TopK(1), 8 new tokens; 5 modes."""
    traces = []
    g = torch.Generator().manual_seed(7)
    for dt in ("fp32", "fp16", "bf16"):
        for mode in ("plain", "cd", "dd", "dd_unk", "both"):
            rng = np.random.default_rng(1234)
            n_q = 32 if (mode == "dd" and dt == "fp16") else 4
            for q in range(n_q):
                ids = pope_like_ids(rng, vocab=97)
                img = torch.randn(1, 3, 2, 2, generator=g)
                img_cd = img * 0.5 + torch.randn(1, 3, 2, 2, generator=g)
                model = ToyVLM(logit_dtype=DTYPES[dt])
                out = run_reference(model, ids.clone(), max_length=ids.shape[1] + 8, warp={"top_k": 1},
                                    multinomial=ARGMAX_MN, **loop_kwargs(mode, ids, img, img_cd))
                traces.append({"dtype": dt, "mode": mode, "q": q, "ids": ids.tolist(), "img": img.tolist(),
                               "img_cd": img_cd.tolist(), "tokens": out.sequences[:, ids.shape[1]:].tolist(),
                               "schedule": [list(map(lambda x: list(x) if isinstance(x, tuple) else x, c)) for c in model.calls],
                               "score_sha256": [hashlib.sha256(to_bits(s).tobytes()).hexdigest() for s in out.scores]})
    with open(os.path.join(HERE, "loop_traces.json"), "w") as f:
        json.dump(traces, f)
    print("loop traces:", len(traces))


def gen_eos_pad():
    """vcd_sample.py:257-299: pad after EOS, multi-EOS product, stop when all rows finished."""
    cases = []
    V = 50
    for eos, plan in (([2], [[7, 2, 9, 9, 9], [7, 8, 9, 2, 9], [7, 8, 9, 10, 11]]),
                      ([2, 5], [[5, 9, 9, 9, 9], [7, 2, 9, 9, 9], [7, 8, 2, 9, 9]]),
                      (2, [[7, 2, 9, 9, 9]])):
        plan_t = torch.tensor(plan)
        B, S = plan_t.shape
        bank = []
        for s in range(S):
            for _branch in range(2):
                row = torch.zeros(B, V, dtype=torch.float16)
                row[torch.arange(B), plan_t[:, s]] = 9.0
                bank.append(row)
        ids = torch.ones(B, 4, dtype=torch.long)
        out = run_reference(BankModel(bank), ids, max_length=4 + S, warp={"top_k": 1}, pad=0, eos=eos,
                            multinomial=ARGMAX_MN, attention_mask=torch.ones_like(ids), use_dd_unk=True,
                            cd_alpha=1.0, cd_beta=0.1)
        cases.append({"eos": eos, "pad": 0, "plan": plan, "V": V, "sequences": out.sequences.tolist(),
                      "n_scores": len(out.scores)})
    # eos without pad -> ValueError (vcd_sample.py:258-259)
    try:
        run_reference(BankModel(bank, pad=None), ids, max_length=6, warp={"top_k": 1}, pad=None, eos=2,
                      multinomial=ARGMAX_MN, attention_mask=torch.ones_like(ids), use_dd_unk=True)
        err = None
    except ValueError as e:
        err = str(e)
    with open(os.path.join(HERE, "eos_pad.json"), "w") as f:
        json.dump({"cases": cases, "eos_without_pad_error": err}, f)
    print("eos/pad:", len(cases))


def gen_noise():
    ref = load_reference()
    arrays = {}
    for t in (0, 1, 500, 999):
        torch.manual_seed(100 + t)
        x = torch.randn(3, 6, 5)
        torch.manual_seed(200 + t)
        y = ref.add_diffusion_noise(x, t)
        arrays[f"x_{t}"] = x.numpy()
        arrays[f"y_{t}"] = y.numpy()
    np.savez(os.path.join(HERE, "noise.npz"), **arrays)
    print("noise: ok")


class FakeTok:
    """token id -> string with case/space collisions, for metrics.calibrate_label_dict."""
    TABLE = {0: "<unk>", 1: "Yes", 2: " yes", 3: "No", 4: "no ", 5: "YES", 6: "maybe"}

    def decode(self, i):
        return self.TABLE.get(int(i), f"t{int(i)}")


def gen_calibration():
    ref = load_reference()
    out = {"label_dict": [], "affine": []}
    rng = np.random.default_rng(5)
    for dt in ("fp16", "bf16", "fp32"):
        for trial in range(3):
            row = torch.from_numpy(rng.standard_normal((1, 40), dtype=np.float32) * 3).to(DTYPES[dt])
            row[0, rng.integers(0, 7)] += 6
            d = ref.metrics.calibrate_label_dict(row, FakeTok())
            p = ref.metrics.get_prob_from_logits(d)
            out["label_dict"].append({"dtype": dt, "row_bits": to_bits(row).tolist(), "dict": d, "p": p})
    for mode in ("diagonal_W", "identity_W"):
        probs = rng.random((6, 2))
        p_cf = rng.random(2) * 0.5 + 0.1
        labels = rng.integers(0, 2, size=6)
        acc, cal = ref.metrics.eval_accuracy(probs, labels, mode=mode, p_cf=p_cf)
        out["affine"].append({"mode": mode, "probs": probs.tolist(), "p_cf": p_cf.tolist(), "labels": labels.tolist(),
                              "acc": float(acc), "calibrated": [c.reshape(-1).tolist() for c in cal]})
    acc, cal = ref.metrics.eval_accuracy(probs, labels)
    out["affine"].append({"mode": None, "probs": probs.tolist(), "p_cf": None, "labels": labels.tolist(),
                          "acc": float(acc), "calibrated": [c.reshape(-1).tolist() for c in cal]})
    with open(os.path.join(HERE, "calibration.json"), "w") as f:
        json.dump(out, f)
    print("calibration: ok")


def gen_scorers():
    """Drive the reference's POPE scorer SCRIPTS (eval_pope.py, eval_pope_calibrate.py) on synthetic answer files and
    record what they print.  The calibrate script hard-codes relative paths, so it runs in a scratch cwd."""
    import re
    import subprocess
    import tempfile
    from ref_shim import REF_ROOT
    rng = np.random.default_rng(11)
    n = 60
    words = ["Yes", "No", "yes, it is", "no.", "Maybe", "The answer is no", "YES"]
    gt, gen = [], []
    for i in range(n):
        lab = "yes" if rng.random() < 0.5 else "no"
        def td():
            py = float(rng.random()); pn = float(rng.random())
            d = {"Yes" if rng.random() < 0.5 else "yes": py, "no": pn, "maybe": 0.01}
            if rng.random() < 0.2:
                d.pop("no")
            return d
        gt.append({"question_id": i, "label": lab, "text": "q", "image": "x.jpg"})
        gen.append({"question_id": i, "text": words[int(rng.integers(len(words)))], "naive": td(), "unk": td(), "none": td()})
    out = {"gt": gt, "gen": gen}
    with tempfile.TemporaryDirectory() as d:
        gtp, gp = os.path.join(d, "gt.json"), os.path.join(d, "gen.jsonl")
        open(gtp, "w").write("\n".join(json.dumps(x) for x in gt))
        open(gp, "w").write("\n".join(json.dumps(x) for x in gen))
        r = subprocess.run([sys.executable, os.path.join(REF_ROOT, "experiments/eval/eval_pope.py"), "--gt_files", gtp, "--gen_files", gp],
                           capture_output=True, text=True, check=True)
        out["eval_pope"] = {k.lower(): float(v) for k, v in re.findall(r"^(Precision|Recall|F1|Accuracy|yes|unknow): ([0-9.eE+-]+)", r.stdout, flags=re.M)}
        for split in ("random", "popular", "adversarial"):
            os.makedirs(os.path.join(d, "experiments/data/POPE/gqa"), exist_ok=True)
            os.makedirs(os.path.join(d, "experiments/output/llava-13B"), exist_ok=True)
            open(os.path.join(d, f"experiments/data/POPE/gqa/gqa_pope_{split}.json"), "w").write("\n".join(json.dumps(x) for x in gt))
            open(os.path.join(d, f"experiments/output/llava-13B/llava_gqa_pope_{split}_seed55_both.jsonl"), "w").write("\n".join(json.dumps(x) for x in gen))
        r = subprocess.run([sys.executable, os.path.join(REF_ROOT, "experiments/eval/eval_pope_calibrate.py")], cwd=d,
                           capture_output=True, text=True, check=True, env={**os.environ, "PYTHONDONTWRITEBYTECODE": "1"})
        blocks = re.findall(r"Evaluate the performance in (\w+) setting\nF1: ([0-9.]+) Accuracy: ([0-9.]+) Precision: ([0-9.]+) \t Recall: ([0-9.]+) \t yes: ([0-9.]+) unknow: ([0-9.]+) number questions (\d+) confidence ([0-9.eE+-]+)", r.stdout)
        out["eval_pope_calibrate"] = {b[0]: {"f1": float(b[1]), "accuracy": float(b[2]), "precision": float(b[3]), "recall": float(b[4]),
                                             "yes": float(b[5]), "n": int(b[7]), "confidence": float(b[8])} for b in blocks[:4]}
    with open(os.path.join(HERE, "scorers.json"), "w") as f:
        json.dump(out, f)
    print("scorers:", out["eval_pope"], list(out["eval_pope_calibrate"]))


def gen_scorers_all():
    """eval_pope_calibrate.py over answer files that carry all five label dicts (naive, noise, none, zero, unk: what qwen_calibrate.py
    writes) with the setting list of its line 82 - the one the script keeps as a comment, ['naive', 'noise', 'none', 'zero', 'unk',
    'none_noise', 'none_unk', 'none_unk_noise', 'all'] - in place of line 83's four.  The script is read, that ONE line is swapped in
    memory, and the result runs from a scratch directory (PYTHONPATH = the reference's experiments/ for its utils.metrics import)."""
    import re
    import subprocess
    import tempfile
    from ref_shim import REF_ROOT
    rng = np.random.default_rng(29)
    n = 80
    gt, gen = [], []
    for i in range(n):
        lab = "yes" if rng.random() < 0.5 else "no"
        def td():
            py = float(rng.random()); pn = float(rng.random())
            d = {"Yes" if rng.random() < 0.5 else "yes": py, "no": pn, "maybe": 0.01}
            if rng.random() < 0.15:
                d.pop("no")
            return d
        gt.append({"question_id": i, "label": lab, "text": "q", "image": "x.jpg"})
        gen.append({"question_id": i, "text": "yes", "naive": td(), "noise": td(), "none": td(), "zero": td(), "unk": td()})
    out = {"gt": gt, "gen": gen}
    src = open(os.path.join(REF_ROOT, "experiments/eval/eval_pope_calibrate.py")).read()
    four = "    for name in ['naive', 'none', 'unk', 'none_unk']:"
    assert src.count(four) == 1
    names = ["naive", "noise", "none", "zero", "unk", "none_noise", "none_unk", "none_unk_noise", "all"]
    src = src.replace(four, "    for name in %r:" % names)
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "scorer_all_settings.py"), "w").write(src)
        for split in ("random", "popular", "adversarial"):
            os.makedirs(os.path.join(d, "experiments/data/POPE/gqa"), exist_ok=True)
            os.makedirs(os.path.join(d, "experiments/output/llava-13B"), exist_ok=True)
            open(os.path.join(d, f"experiments/data/POPE/gqa/gqa_pope_{split}.json"), "w").write("\n".join(json.dumps(x) for x in gt))
            open(os.path.join(d, f"experiments/output/llava-13B/llava_gqa_pope_{split}_seed55_both.jsonl"), "w").write("\n".join(json.dumps(x) for x in gen))
        r = subprocess.run([sys.executable, os.path.join(d, "scorer_all_settings.py")], cwd=d, capture_output=True, text=True, check=True,
                           env={**os.environ, "PYTHONDONTWRITEBYTECODE": "1", "PYTHONPATH": os.path.join(REF_ROOT, "experiments")})
        blocks = re.findall(r"Evaluate the performance in (\w+) setting\nF1: ([0-9.]+) Accuracy: ([0-9.]+) Precision: ([0-9.]+) \t Recall: ([0-9.]+) \t yes: ([0-9.]+) unknow: ([0-9.]+) number questions (\d+) confidence ([0-9.eE+-]+)", r.stdout)
        out["eval_pope_calibrate"] = {b[0]: {"f1": float(b[1]), "accuracy": float(b[2]), "precision": float(b[3]), "recall": float(b[4]),
                                             "yes": float(b[5]), "n": int(b[7]), "confidence": float(b[8])} for b in blocks[:len(names)]}
        assert list(out["eval_pope_calibrate"]) == names
    with open(os.path.join(HERE, "scorers_all.json"), "w") as f:
        json.dump(out, f)
    print("scorers_all:", {k: v["accuracy"] for k, v in out["eval_pope_calibrate"].items()})


def gen_mme_convert():
    """Runs the reference's MME converter SCRIPT (experiments/eval/MME/convert_answer_to_mme_calibrate.py) unmodified on a synthetic
    benchmark tree + answers file and records what it writes.  The script hard-codes an absolute ground-truth path
    (/mnt/data/xue.w/yf/data/MME_Benchmark) and cwd-relative answer paths, so it runs in a scratch cwd inside a wrapper process
    that only remaps that path prefix to the scratch tree for os.listdir / os.path.isdir / os.path.exists / open."""
    import subprocess
    import tempfile
    from ref_shim import REF_ROOT
    rng = np.random.default_rng(23)
    cats = ["existence", "count", "position", "color", "commonsense_reasoning", "numerical_calculation", "text_translation", "code_reasoning"]
    gt_rows, answers = [], []
    words = ["Yes", "No", "yes", "no.", "Maybe", "Yes, it is"]
    for ci, cat in enumerate(cats):
        for img in range(3):
            ext = "png" if ci % 2 else "jpg"
            for qi in range(2):
                base_q = f"Is there item {img}-{qi} in the {cat} picture?"
                double = (img + qi + ci) % 3 == 0                      # some ground-truth questions carry the double space
                gt_q = base_q + ("  " if double else " ") + "Please answer yes or no."
                gt_rows.append((cat, f"{img:04d}.txt", gt_q, "Yes" if rng.random() < 0.5 else "No", ci % 2 == 0))

                def td():
                    d = {"yes": float(rng.random()), "no": float(rng.random()), "maybe": 0.01}
                    if rng.random() < 0.15:
                        d.pop("no")
                    return d
                prompt = base_q + "\nAnswer the question using a single word or phrase." if (img + qi) % 2 == 0 else gt_q
                answers.append({"question_id": f"{cat}/{img:04d}.{ext}", "prompt": prompt, "text": words[int(rng.integers(len(words)))],
                                "naive": td(), "none": td(), "unk": td()})
    out = {"gt": [list(r[:4]) for r in gt_rows], "answers": answers, "results": {}}
    with tempfile.TemporaryDirectory() as d:
        bench = os.path.join(d, "MME_Benchmark")
        for cat, file, q, a, nested in gt_rows:
            qa = os.path.join(bench, cat, "questions_answers_YN") if nested else os.path.join(bench, cat)
            os.makedirs(qa, exist_ok=True)
            if nested:
                os.makedirs(os.path.join(bench, cat, "images"), exist_ok=True)
            with open(os.path.join(qa, file), "a") as f:
                f.write(q + "\t" + a + "\n")
        os.makedirs(os.path.join(d, "eval/MME/answers/MME_sft_dd"))
        open(os.path.join(d, "eval/MME/answers/MME_sft_dd/exp.jsonl"), "w").write("\n".join(json.dumps(x) for x in answers))
        script = os.path.join(REF_ROOT, "experiments/eval/MME/convert_answer_to_mme_calibrate.py")
        wrapper = (
            "import os, sys, builtins, runpy\n"
            "PRE, REAL = '/mnt/data/xue.w/yf/data/MME_Benchmark', sys.argv[1]\n"
            "m = lambda p: REAL + p[len(PRE):] if isinstance(p, str) and p.startswith(PRE) else p\n"
            "for mod, name in ((os, 'listdir'), (os.path, 'isdir'), (os.path, 'exists'), (builtins, 'open')):\n"
            "    f = getattr(mod, name)\n"
            "    setattr(mod, name, (lambda f: lambda p, *a, **k: f(m(p), *a, **k))(f))\n"
            "sys.dont_write_bytecode = True\n"
            "sys.argv = [sys.argv[2], '--experiment', 'exp']\n"
            "runpy.run_path(sys.argv[0], run_name='__main__')\n")
        subprocess.run([sys.executable, "-c", wrapper, bench, script], cwd=d, check=True, capture_output=True, text=True,
                       env={**os.environ, "PYTHONDONTWRITEBYTECODE": "1"})
        for name in ("naive", "none", "unk", "none_unk"):
            rd = os.path.join(d, "eval/MME/eval_tool/answers", f"exp-{name}")
            out["results"][name] = {f[:-4]: open(os.path.join(rd, f)).read().splitlines() for f in sorted(os.listdir(rd))}
    with open(os.path.join(HERE, "mme_convert.json"), "w") as f:
        json.dump(out, f)
    print("mme_convert:", {k: sum(len(v) for v in r.values()) for k, r in out["results"].items()})


PROC_SPECS = [
    # (name, spec); spec keys: min_new (n, eos list), min_len (n, eos list), rep (penalty), stop (True: sequences taken from a
    # processor-free run of the same case so that they really occur)
    ("min_new", {"min_new": 2}),
    ("min_len", {"min_len": 6}),
    ("rep", {"rep": 1.3}),
    ("stop", {"stop": True}),
    ("qwen_mme", {"min_new": 1, "stop": True}),
    ("all", {"rep": 1.5, "min_len": 5, "min_new": 3, "stop": True}),
]


def build_hf_processors(spec, prompt_len, eos, stop_words):
    """The list HF's generate() would hand to sample() for these kwargs, in HF's order (repetition penalty, min_length,
    min_new_tokens, then the caller's own = Qwen's stop-words processor, modeling_qwen.py:1061-1075), made of HF's classes and the
    reference's own StopWordsLogitsProcessor."""
    from transformers.generation.logits_process import (LogitsProcessorList, MinLengthLogitsProcessor,
                                                        MinNewTokensLengthLogitsProcessor, RepetitionPenaltyLogitsProcessor)
    from ref_shim import load_qwen_stop_words
    lst = LogitsProcessorList()
    if "rep" in spec:
        lst.append(RepetitionPenaltyLogitsProcessor(penalty=spec["rep"]))
    if "min_len" in spec:
        lst.append(MinLengthLogitsProcessor(spec["min_len"], eos))
    if "min_new" in spec:
        lst.append(MinNewTokensLengthLogitsProcessor(prompt_len, spec["min_new"], eos))
    if spec.get("stop"):
        lst.append(load_qwen_stop_words()(stop_words_ids=stop_words, eos_token_id=eos[0]))
    return lst


def gen_processors():
    """Real reference sample() with a `logits_processor` list between contrast and warp (vcd_sample.py:197 / :204)."""
    cases = []
    arrays = {}
    steps, B, L0 = 5, 3, 4
    for V in (97, 1003):
        for dt in ("fp32", "fp16", "bf16"):
            for n_in in (1, 2, 3):
                if V == 1003 and n_in == 3:
                    continue
                for name, spec in PROC_SPECS:
                    for warp in ({"top_k": 1}, {"temperature": 0.7, "top_k": 1}):
                        if warp.get("temperature") and name not in ("qwen_mme", "all"):
                            continue
                        seed = 5000 + len(cases)
                        rows = logit_rows(seed, B, V, DTYPES[dt], n_in, "normal", steps)
                        bank = [r for step in rows for r in step]
                        ids = torch.ones(B, L0, dtype=torch.long)
                        if "rep" not in spec:           # HF's repetition penalty gathers scores[input_ids]: an image slot (-200) is out of
                            ids[:, 2] = -200            # range, so only Qwen / InstructBLIP-style prompts (no slot) ever combine with it
                        ids[:, 3] = torch.tensor([5, 6, 7])
                        kw = dict(attention_mask=torch.ones_like(ids), cd_alpha=1.0, cd_beta=0.1, **MODE_KW[n_in])
                        free = run_reference(BankModel([b.clone() for b in bank]), ids.clone(), max_length=L0 + steps, warp=warp,
                                             multinomial=ARGMAX_MN, **kw)
                        toks = free.sequences[:, L0:].tolist()
                        eos = [toks[0][0], toks[2][1]]         # ids the free run really emits early: the EOS masks bite
                        # stop sequences that occur: row 0's first two tokens, row 1's token at step 2, one ending in the prompt's
                        # last id (matches at step 0 from the PROMPT tail), and [eos] itself (dropped by the constructor)
                        stop_words = [toks[0][:2], [toks[1][2]], [6], [eos[0]]]
                        procs = build_hf_processors(spec, L0, eos, stop_words)
                        out = run_reference(BankModel([b.clone() for b in bank]), ids.clone(), max_length=L0 + steps, warp=warp,
                                            pad=0, eos=eos, multinomial=ARGMAX_MN, logits_processor=procs, **kw)
                        ci = len(cases)
                        for s_, sc in enumerate(out.scores):
                            arrays[f"p{ci}_s{s_}"] = to_bits(sc)
                        cases.append({"id": ci, "V": V, "dtype": dt, "n_in": n_in, "proc": name, "spec": spec, "warp": warp, "seed": seed,
                                      "steps": steps, "B": B, "ids": ids.tolist(), "eos": eos, "pad": 0, "stop_words": stop_words,
                                      "free_tokens": toks, "sequences": out.sequences.tolist(), "n_scores": len(out.scores)})
    np.savez_compressed(os.path.join(HERE, "processors.npz"), **arrays)
    with open(os.path.join(HERE, "processors.json"), "w") as f:
        json.dump(cases, f)
    changed = sum(c["sequences"] != [r[:len(c["sequences"][0])] for r in [i + t for i, t in zip(c["ids"], c["free_tokens"])]] for c in cases)
    print("processors:", len(cases), "cases;", changed, "differ from the processor-free run")


if __name__ == "__main__":
    if len(sys.argv) > 1:                       # python make_golden.py processors ...: regenerate only the named sets
        torch.set_num_threads(8)
        for name in sys.argv[1:]:
            globals()["gen_" + name]()
        sys.exit(0)
    torch.set_num_threads(8)
    gen_kernel_vectors()
    gen_loop_traces()
    gen_eos_pad()
    gen_noise()
    gen_calibration()
    gen_scorers()
    gen_scorers_all()
    gen_processors()
    gen_mme_convert()
    gen_kernel_vectors_gpu_scalar()
