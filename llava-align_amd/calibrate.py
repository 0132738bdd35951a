"""Post-hoc calibration and POPE scoring — the consumers of the decoding path's outputs
(SURVEY.md §8 f.1, f.4).  Host-side float64 arithmetic exactly as the reference does it; the only
device work (softmax of the step-0 scores row and its top-k) already happened inside the fused
sampling kernel (`top_prob`, `top_tok`).

  label_dict_from_top      experiments/utils/metrics.py:102-113  calibrate_label_dict
  get_prob_from_logits     experiments/utils/metrics.py:115-125 (and eval_pope_calibrate.py:18-29)
  affine_calibrate         experiments/eval/eval_pope_calibrate.py:65-74,136-141 (metrics.py:8-41)
  pope_scores              experiments/eval/eval_pope.py:27-67
  pope_scores_calibrated   experiments/eval/eval_pope_calibrate.py:84-180 ('individual' mode)
  AnswerWriter             experiments/eval/calibrate/llava_calibrate.py:209-219 (JSONL schema, flush per question)
  mme_convert / write_mme_results   experiments/eval/MME/convert_answer_to_mme_calibrate.py:59-139 (and convert_answer_to_mme.py:52-70)
  mme_scores               experiments/eval/MME/eval_tool/calculation.py
"""
from __future__ import annotations

import json
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import numpy as np

LABEL_DICT = {0: ["yes"], 1: ["no"]}
LABEL_TO_INT = {"yes": 0, "no": 1}


def label_dict_from_top(top_tok: Sequence[int], top_prob: Sequence[float], decode: Callable[[int], str]) -> Dict[str, float]:
    """Top-k (token, prob) pairs in descending probability -> {normalised string: prob}; the FIRST (highest)
    occurrence of a string wins, later collisions are dropped, not summed."""
    out: Dict[str, float] = {}
    for tok, p in zip(top_tok, top_prob):
        if int(tok) < 0:
            continue
        s = decode(int(tok)).lower().strip()
        if s not in out:
            out[s] = float(p)
    return out


def get_prob_from_logits(top_token_probs: Dict[str, float], label_dict=LABEL_DICT) -> List[float]:
    probs = {k.lower().strip(): v for k, v in top_token_probs.items()}
    return [sum(probs.get(a.lower(), 0) for a in answers) for _, answers in label_dict.items()]


def calibrate_weight(p_cf, mode: str = "diagonal_W"):
    n = len(p_cf)
    if mode == "diagonal_W":
        return np.linalg.inv(np.identity(n) * p_cf), np.zeros([n, 1])
    if mode == "identity_W":
        return np.identity(n), -1 * np.expand_dims(p_cf, axis=-1)
    raise AssertionError(mode)


def affine_calibrate(p, p_cf=None, mode: str = "diagonal_W", eps: float = 1e-4):
    """q = W p + b renormalised; p_cf (content-free prior) is normalised and offset by eps as the POPE scorer does."""
    p = np.asarray(p, dtype=np.float64)
    p = p / np.sum(p)
    n = p.shape[0]
    if p_cf is None:
        W, b = np.identity(n), np.zeros([n, 1])
    else:
        p_cf = np.asarray(p_cf, dtype=np.float64)
        p_cf = p_cf / np.sum(p_cf)
        W, b = calibrate_weight([x + eps for x in p_cf], mode)
    q = np.matmul(W, np.expand_dims(p, axis=-1)) + b
    q /= np.sum(q)
    return q.reshape(-1), int(np.argmax(q))


def _prf(tp, tn, fp, fn, yes, unknown, total):
    precision = tp / (tp + fp)
    recall = tp / (tp + fn)
    return {"precision": precision, "recall": recall, "f1": 2 * precision * recall / (precision + recall),
            "accuracy": (tp + tn) / total, "yes": yes / total, "unknown": unknown / total, "n": total}


def pope_scores(gt: Sequence[dict], gen: Sequence[dict]) -> dict:
    """String-match scoring of generated answers: 'yes' / 'no' substring tests, exactly eval_pope.py."""
    tp = tn = fp = fn = unknown = yes = 0
    for g, a in zip(gt, gen):
        assert g["question_id"] == a["question_id"]
        label, text = g["label"].lower().strip(), a["text"].lower().strip()
        if label == "yes":
            if "yes" in text:
                tp += 1; yes += 1
            else:
                fn += 1
        elif label == "no":
            if "no" in text:
                tn += 1
            else:
                yes += 1; fp += 1
        else:
            unknown += 1
    return _prf(tp, tn, fp, fn, yes, unknown, len(gt))


CALIBRATE_SOURCES = {"none_noise": ("noise", "none"), "none_unk": ("unk", "none"), "none_unk_noise": ("noise", "none", "unk"),
                     "all": ("noise", "none", "zero", "unk")}          # eval_pope_calibrate.py:119-130: priors that are summed


def calibrate_sources(name: str) -> tuple:
    """The answer-file keys a calibration setting reads: 'naive' needs no prior, a plain name reads itself, the combined settings read the
    priors they sum."""
    return ("naive",) if name == "naive" else ("naive",) + CALIBRATE_SOURCES.get(name, (name,))


def _calibrated_row(p, cf, mode: str):
    """One question, as the script computes it (eval_pope_calibrate.py:115-139): p, cf = the raw label probabilities of the answer and of the
    prior (None: no prior).  -> q [2, 1], possibly NaN (neither label among a top-10: 0 / 0, no guard in the reference)."""
    with np.errstate(invalid="ignore", divide="ignore"):
        p = p / np.sum(p)
        W, b = np.identity(2), np.zeros([2, 1])
        if cf is not None:
            cf = cf / np.sum(cf)
            W, b = calibrate_weight([x + 1e-4 for x in cf], mode)
        q = np.matmul(W, np.expand_dims(p, axis=-1)) + b
        q /= np.sum(q)
    return q


def _calibrated_rows(P: np.ndarray, CF: Optional[np.ndarray], mode: str) -> np.ndarray:
    """All questions at once, bit for bit the rows of `_calibrated_row` (tests/test_calibrate.py compares them): 3,000 questions x 4 settings
    were 0.2 - 0.6 s of numpy calls on 2-vectors.  With W = inv(diag(c)) = diag(1 / c) exactly and b = 0, q = (p / c) renormalised; rows that
    are not finite everywhere (and the identity_W mode) take the per-row path, whose NaN pattern decides their arg-max."""
    n = P.shape[0]
    with np.errstate(invalid="ignore", divide="ignore"):
        p = P / (P[:, :1] + P[:, 1:])
        if CF is None:
            q = p.copy()
        else:
            c = CF / (CF[:, :1] + CF[:, 1:]) + 1e-4
            q = (1.0 / c) * p
        q = q / (q[:, :1] + q[:, 1:])
    slow = ~np.isfinite(q).all(1) | ~np.isfinite(p).all(1)
    if CF is not None:
        slow |= ~np.isfinite(c).all(1)
    if mode != "diagonal_W":
        slow[:] = True
    for i in np.nonzero(slow)[0]:
        q[i] = _calibrated_row(P[i], None if CF is None else CF[i], mode).reshape(-1)
    return q


def pope_scores_calibrated(gt: Sequence[dict], gen: Sequence[dict], name: str = "naive", mode: str = "diagonal_W") -> dict:
    """'individual' calibration per question (eval_pope_calibrate.py:115-139): p from gen['naive'], prior from gen[name] ('none', 'unk',
    'noise', 'zero' / 'zeros', ...) or the sum of several ('none_unk', 'none_noise', 'none_unk_noise', 'all': CALIBRATE_SOURCES); arg-max of
    the calibrated 2-vector is the answer (0 = yes)."""
    for g, a in zip(gt, gen):
        assert g["question_id"] == a["question_id"]
    n = len(gt)
    labels = [LABEL_TO_INT[g["label"]] for g in gt]
    P = np.array([get_prob_from_logits(a["naive"]) for a in gen], dtype=np.float64).reshape(n, 2)
    CF = None
    if name != "naive":
        srcs = CALIBRATE_SOURCES.get(name, (name,))
        CF = np.array([get_prob_from_logits(a[srcs[0]]) for a in gen], dtype=np.float64).reshape(n, 2)
        for src in srcs[1:]:                                   # summed in the script's order
            CF = CF + np.array([get_prob_from_logits(a[src]) for a in gen], dtype=np.float64).reshape(n, 2)
    q = _calibrated_rows(P, CF, mode)
    # neither label among the top-10 of the answer (or of the prior): 0 / 0 -> NaN, whose arg-max is class 0 = "yes" in the reference
    # (eval_pope_calibrate.py:65-74 has no guard).  Scored the same way here, but counted: `nan_rows` of the result.
    nan_rows = int((~np.isfinite(q).all(1)).sum())
    ans = np.argmax(q, axis=1).tolist()                        # (the first NaN of a row, like np.argmax of the row alone)
    confidence = 0.0
    for v in np.max(q, axis=1).tolist():                       # accumulated in question order, as the script does
        confidence += v
    tp = tn = fp = fn = unknown = yes = 0
    for label, a_ in zip(labels, ans):
        if label == 0:
            if a_ == 0:
                tp += 1; yes += 1
            else:
                fn += 1
        else:
            if a_ == 1:
                tn += 1
            else:
                yes += 1; fp += 1
    out = _prf(tp, tn, fp, fn, yes, unknown, n)
    out["confidence"] = confidence / n
    out["nan_rows"] = nan_rows
    return out


class AnswerWriter:
    """One JSON line per question, flushed immediately (the reference's only crash tolerance)."""
    FIELDS = ("question_id", "prompt", "text", "model_id", "image", "logits_score", "naive", "unk", "none", "metadata")

    def __init__(self, path: str):
        self.f = open(path, "w")

    def write(self, question_id, prompt, text, model_id, image, logits_score, naive, unk=None, none=None, metadata=None, extra=None):
        """extra: further label dicts (image priors 'noise' / 'zeros' / 'ones', test_samples_llava.py:148-158), written before metadata."""
        rec = {"question_id": question_id, "prompt": prompt, "text": text, "model_id": model_id, "image": image,
               "logits_score": logits_score, "naive": naive, "unk": unk, "none": none}
        rec.update(extra or {})
        rec["metadata"] = metadata or {}
        self.f.write(json.dumps(rec) + "\n")
        self.f.flush()

    def close(self):
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


# ------------------------------------------------------------------ MME scoring (experiments/eval/MME/eval_tool/calculation.py)
MME_TASKS = {"Perception": ["existence", "count", "position", "color"],
             "Cognition": ["commonsense_reasoning", "numerical_calculation", "text_translation", "code_reasoning"]}


def mme_parse_pred(pred: str) -> str:
    """calculation.py:22-36 — exact 'yes'/'no', else look for them in the first 4 characters, else 'other'."""
    if pred in ("yes", "no"):
        return pred
    head = pred[:4]
    return "yes" if "yes" in head else ("no" if "no" in head else "other")


def mme_task_score(lines: Sequence[str]) -> dict:
    """One task file: lines 'image<TAB>question<TAB>gt<TAB>prediction', two consecutive questions per image.
    score = 100 * accuracy + 100 * accuracy+ (both questions of an image right), calculation.py:95-140."""
    assert len(lines) % 2 == 0
    hits, plus, other = 0, 0, 0
    for i in range(0, len(lines), 2):
        ok = 0
        for item in lines[i:i + 2]:
            _img, _q, gt, pred = item.split("\t")
            gt, pred = gt.lower(), mme_parse_pred(pred.lower())
            assert gt in ("yes", "no")
            ok += int(gt == pred)
            other += int(pred == "other")
        hits += ok
        plus += int(ok == 2)
    acc, acc_plus = hits / len(lines), plus / (len(lines) // 2)
    return {"acc": acc, "acc_plus": acc_plus, "other_num": other, "score": acc * 100 + acc_plus * 100}


def mme_scores(results_dir: str) -> dict:
    import os
    out = {}
    for group, tasks in MME_TASKS.items():
        per = {t: mme_task_score(open(os.path.join(results_dir, t + ".txt")).readlines()) for t in tasks}
        out[group] = {"total": sum(v["score"] for v in per.values()), "tasks": {t: v["score"] for t, v in per.items()}}
    return out


# ------------------------------------------------------------------ MME answers -> per-task result files
MME_CALIBRATE_NAMES = ("naive", "none", "unk", "none_unk")


def mme_gt_key_prompt(category: str, file: str, prompt: str, gt: Dict[tuple, str]) -> str:
    """The question string under which the benchmark's ground truth lists this answer (convert_answer_to_mme_calibrate.py:128-133):
    the eval prompt's 'Answer the question using a single word or phrase.' is dropped, ' Please answer yes or no.' appended, with a
    DOUBLE space where the single-space form is not a ground-truth key."""
    if "Answer the question using a single word or phrase." in prompt:
        prompt = prompt.replace("Answer the question using a single word or phrase.", "").strip()
    if "Please answer yes or no." not in prompt:
        prompt = prompt + " Please answer yes or no."
        if (category, file, prompt) not in gt:
            prompt = prompt.replace(" Please answer yes or no.", "  Please answer yes or no.")
    return prompt


def mme_convert(answers: Sequence[dict], gt: Dict[tuple, str], names: Sequence[str] = MME_CALIBRATE_NAMES,
                calibrate_mode: str = "individual", mode: str = "diagonal_W") -> Dict[str, Dict[str, List[str]]]:
    """MME answers (JSONL records with question_id 'category/image.ext', prompt, text, naive / none / unk label dicts) ->
    {name: {category: ['file<TAB>question<TAB>gt<TAB>answer', ...]}}: what the reference writes to
    eval_tool/answers/<experiment>-<name>/<category>.txt.  'naive' keeps the generated text; the other names answer 'Yes' / 'No'
    by the arg-max of the affine-calibrated label probabilities, the prior taken per question ('individual': p_cf = prior + 1e-4,
    NOT renormalised, :113-114) or as the mean over the whole file ('all', :79-96).  gt: {(category, file, question): answer}
    (get_gt, :22-41)."""
    label = {0: "yes", 1: "no"}
    prob = {n: [get_prob_from_logits(a[n]) for a in answers] for n in ("naive", "none", "unk")}
    out: Dict[str, Dict[str, List[str]]] = {}
    for name in names:
        res: Dict[str, List[tuple]] = {}
        W, b = np.identity(2), np.zeros([2, 1])
        if calibrate_mode == "all" and name != "naive":
            all_p = np.array(prob["unk"]) + np.array(prob["none"]) if name == "none_unk" else np.array(prob[name])
            p_cf = np.mean(all_p, axis=0)
            W, b = calibrate_weight(p_cf / np.sum(p_cf), mode)
        for i, a in enumerate(answers):
            category = a["question_id"].split("/")[0]
            file = a["question_id"].split("/")[-1].split(".")[0] + ".txt"
            if name == "naive":
                res.setdefault(category, []).append((file, a["prompt"], a["text"]))
                continue
            with np.errstate(invalid="ignore", divide="ignore"):     # neither label among a top-10: 0 / 0 -> NaN -> arg-max 0, as the script computes it
                if calibrate_mode == "individual":
                    if name == "none_unk":
                        s_ = np.array(prob["unk"][i]) + np.array(prob["none"][i])
                        p_cf = s_ / np.sum(s_)
                    else:
                        p_cf = prob[name][i]
                    W, b = calibrate_weight([x + 1e-4 for x in p_cf], mode)
                q = np.matmul(W, np.expand_dims(prob["naive"][i], axis=-1)) + b
                q /= np.sum(q)
            res.setdefault(category, []).append((file, a["prompt"], label[int(np.argmax(q))].capitalize()))
        out[name] = {}
        for category, tups in res.items():
            lines = []
            for file, prompt, ans in tups:
                prompt = mme_gt_key_prompt(category, file, prompt, gt)
                lines.append("\t".join((file, prompt, gt[category, file, prompt], ans)))
            out[name][category] = lines
    return out


def write_mme_results(converted: Dict[str, Dict[str, List[str]]], root: str, experiment: str) -> Dict[str, str]:
    """-> {name: directory}; one <category>.txt per task under <root>/<experiment>-<name>/ (the layout calculation.py reads)."""
    import os
    dirs = {}
    for name, cats in converted.items():
        d = os.path.join(root, f"{experiment}-{name}")
        os.makedirs(d, exist_ok=True)
        for category, lines in cats.items():
            with open(os.path.join(d, f"{category}.txt"), "w") as fp:
                fp.write("".join(line + "\n" for line in lines))
        dirs[name] = d
    return dirs


def mme_load_gt(data_path: str) -> Dict[tuple, str]:
    """get_gt (convert_answer_to_mme_calibrate.py:22-41): {(category, file, question): answer} from the MME benchmark tree."""
    import os
    gt = {}
    for category in os.listdir(data_path):
        cdir = os.path.join(data_path, category)
        if not os.path.isdir(cdir):
            continue
        qa = os.path.join(cdir, "questions_answers_YN") if os.path.exists(os.path.join(cdir, "images")) else cdir
        for file in os.listdir(qa):
            if file.endswith(".txt"):
                for line in open(os.path.join(qa, file)):
                    question, answer = line.strip().split("\t")
                    gt[(category, file, question)] = answer
    return gt
