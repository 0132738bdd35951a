"""Command-line entry points of the drivers (the reference runs experiments/eval/calibrate/llava_calibrate.py:222-246,
MME/run_llava.py:253-318, run_qwen.py:240-304 and blip_calibrate.py:113-135 from bash): the checkpoint-directory loader on CPU, and - on the
GPU box - `python -m llava_align_amd.pope_driver` / `mme_driver` end to end from a generated checkpoint directory, alone and under torchrun
with two ranks (gloo, both on the one GPU)."""
import json
import os
import shutil
import subprocess
import sys

import pytest
import torch

from tiny_checkpoint import write_checkpoint

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_checkpoint_directory_is_read_like_the_reference_reads_it(tmp_path):
    from llava_align_amd import checkpoint as K
    from llava_align_amd.engine import preset
    info = write_checkpoint(str(tmp_path))
    cfg = K.config_from_dir(info["ckpt"])
    t = preset("tiny")
    assert cfg.lm == t.lm and cfg.vision == t.vision                            # every shape comes from config.json, none from a preset
    sd = K.load_state_dict(info["ckpt"])
    assert "model.layers.1.mlp.down_proj.weight" in sd and sd["lm_head.weight"].dtype == torch.float16
    tok = K.load_tokenizer(info["ckpt"])
    ids = K.tokenizer_image_token(tok, "w1 w2 <image> yes no")
    assert ids == [1, 6, 7, -200, 3, 4] and tok.unk_token_id == 0 and tok.eos_token_id == 2      # ONE BOS, -200 where <image> stood
    assert K.tokenizer_image_token(tok, "w1 w2") == [1, 6, 7]
    from transformers import CLIPImageProcessor
    x = K.clip_preprocess(CLIPImageProcessor.from_pretrained(info["ckpt"]), os.path.join(info["images"], "im3.png"))
    assert tuple(x.shape) == (3, 56, 56) and x.dtype == torch.float32
    # a release that names its tower on the hub and carries no vision_config: the shapes come from --preset, or the call says what is missing
    c2 = dict(info["config"]); c2.pop("vision_config")
    os.makedirs(tmp_path / "hub", exist_ok=True)
    json.dump(c2, open(tmp_path / "hub" / "config.json", "w"))
    assert K.config_from_dir(str(tmp_path / "hub"), fallback="tiny").vision == t.vision
    with pytest.raises(FileNotFoundError, match="vision-tower"):
        K.config_from_dir(str(tmp_path / "hub"))


def test_mme_cli_sweep_is_the_reference_scripts_sweep():
    import argparse
    from llava_align_amd.mme_driver import _sweep_settings
    a = argparse.Namespace(arch="llava", temperature=0.2, top_p=None, top_k=None, use_dd=False, use_dd_unk=False, no_sweep=False)
    assert _sweep_settings(a) == [("default", 1.0, None, None)]                  # run_llava.py:281-283: temperature 1.0, then exit() without a VDD flag
    a.use_dd_unk = True
    runs = _sweep_settings(a)
    assert len(runs) == 1 + 20 + 21 + 9 and runs[1] == ("temp_0.05", 0.05, None, None) and runs[20][0] == "temp_1.0"
    assert runs[21] == ("top_p_0.0", 0.2, 0.0, None) and runs[41] == ("top_p_1.0", 0.2, 1.0, None) and runs[-1] == ("top_k_500", 0.2, None, 500)
    a.arch, a.temperature = "qwen", 1.0
    assert _sweep_settings(a)[0] == ("default", 1.0, None, None)
    a.no_sweep = True
    assert len(_sweep_settings(a)) == 1


def _env():
    env = dict(os.environ, VDD_FORCE_DEVICE="0", VDD_DIST_BACKEND="gloo", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.gpu
def test_pope_driver_from_a_checkpoint_directory_alone_and_under_torchrun(tmp_path):
    info = write_checkpoint(str(tmp_path))
    common = ["--model-path", info["ckpt"], "--question-file", info["questions"], "--image-folder", info["images"], "--use_dd_unk", "--cd_alpha", "1",
              "--cd_beta", "0.1", "--temperature", "0.5", "--cd_greedy", "--max_new_tokens", "6", "--batch", "6"]
    one = str(tmp_path / "out" / "one.jsonl")
    p1 = subprocess.run([sys.executable, "-m", "llava_align_amd.pope_driver", *common, "--answers-file", one], capture_output=True, text=True, env=_env(),
                        timeout=900, cwd=ROOT)
    assert p1.returncode == 0, p1.stderr[-3000:]
    rep = json.loads(p1.stdout[p1.stdout.index("{"):])
    assert rep["batch_invariant"] is True and rep["world"] == 1 and set(rep["scores"]) == {"string_match", "naive", "none", "unk", "none_unk"}
    two = str(tmp_path / "out" / "two.jsonl")
    p2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                         "29561", "-m", "llava_align_amd.pope_driver", *common, "--answers-file", two], capture_output=True, text=True, env=_env(),
                        timeout=900, cwd=ROOT)
    assert p2.returncode == 0, p2.stderr[-3000:]
    a, b = [json.loads(l) for l in open(one)], [json.loads(l) for l in open(two)]
    assert [x["question_id"] for x in a] == list(range(100, 100 + info["n_questions"]))                 # the file's order, written once by rank 0
    assert tuple(a[0].keys()) == ("question_id", "prompt", "text", "model_id", "image", "logits_score", "naive", "unk", "none", "metadata")
    assert a[0]["model_id"] == "tiny-llava" and all(len(x["text"].split()) >= 1 for x in a)
    assert a == b                                                                                       # 2 ranks == 1 rank, every field
    assert json.loads(p2.stdout[p2.stdout.index("{"):])["world"] == 2


@pytest.mark.gpu
def test_mme_driver_from_a_checkpoint_directory(tmp_path):
    info = write_checkpoint(str(tmp_path))
    cats = ("existence", "count")
    gt_root = tmp_path / "MME"
    qfile = tmp_path / "llava_mme.jsonl"
    with open(qfile, "w") as f:
        for i in range(4):
            cat = cats[i % 2]
            os.makedirs(gt_root / cat, exist_ok=True)
            shutil.copy(os.path.join(info["images"], f"im{i}.png"), gt_root / cat / f"{i:03d}.png")
            lines = []
            for k in range(2):
                q = f"w{i} w{k} ? Please answer yes or no."
                f.write(json.dumps({"question_id": f"{cat}/{i:03d}.png", "image": f"{cat}/{i:03d}.png", "text": q, "category": cat}) + "\n")
                lines.append(f"{q}\t{('Yes', 'No')[(i + k) % 2]}")
            open(gt_root / cat / f"{i:03d}.txt", "w").write("\n".join(lines) + "\n")
    out = str(tmp_path / "answers" / "tiny-setting.jsonl")
    p = subprocess.run([sys.executable, "-m", "llava_align_amd.mme_driver", "--arch", "llava", "--model-path", info["ckpt"], "--question-file", str(qfile),
                        "--image-folder", str(gt_root), "--answers-file", out, "--use_dd_unk", "--max_new_tokens", "4", "--no-sweep", "--gt-root", str(gt_root),
                        "--seed", "1"], capture_output=True, text=True, env=_env(), timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    rep = json.loads(p.stdout.strip().splitlines()[-1])
    assert rep["run"] == "default" and rep["n_answers"] == 8 and rep["answers_file"].endswith("tiny-default.jsonl")
    recs = [json.loads(l) for l in open(rep["answers_file"])]
    assert len(recs) == 8 and set(recs[0]) >= {"question_id", "prompt", "text", "naive", "none", "unk", "answer_id", "model_id"}
    assert os.path.isdir(os.path.join(os.path.dirname(out), "eval_tool_answers"))                      # the converter wrote the scorer's input tree
    # without --no-sweep: the reference scripts' 51 settings (run_llava.py:281-318) in ONE pass over the question file, one answers file per setting
    out2 = str(tmp_path / "sweep" / "tiny-setting.jsonl")
    p2 = subprocess.run([sys.executable, "-m", "llava_align_amd.mme_driver", "--arch", "llava", "--model-path", info["ckpt"], "--question-file", str(qfile),
                         "--image-folder", str(gt_root), "--answers-file", out2, "--use_dd_unk", "--max_new_tokens", "3", "--seed", "1"],
                        capture_output=True, text=True, env=_env(), timeout=900, cwd=ROOT)
    assert p2.returncode == 0, p2.stderr[-3000:]
    runs = [json.loads(l) for l in p2.stdout.strip().splitlines() if l.startswith("{")]
    assert len(runs) == 51 and runs[0]["run"] == "default" and runs[1]["run"] == "temp_0.05" and runs[-1]["run"] == "top_k_500"
    assert all(r["n_answers"] == 8 and os.path.exists(r["answers_file"]) for r in runs) and len({r["answers_file"] for r in runs}) == 51


@pytest.mark.gpu
def test_blip_driver_from_its_parts(tmp_path):
    """`python -m llava_align_amd.blip_driver` (blip_calibrate.py:113-135's arguments): the InstructBLIP state dict under LAVIS's names as one
    safetensors file, the Vicuna directory (here the LM of the generated LLaVA checkpoint + its tokenizer), a generated BERT-style tokenizer
    ([CLS] 101 ... [SEP] 102) for the Q-Former; test-sized towers with the real structure (88-wide ViT heads, cross-attention every 2nd layer)."""
    from safetensors.torch import save_file
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast
    from blip_weights import blip_state_dict
    from llava_align_amd.blip_frontend import tiny_blip_config
    info = write_checkpoint(str(tmp_path))
    bcfg = tiny_blip_config()
    save_file({k: v.to(torch.float16).contiguous() for k, v in blip_state_dict(bcfg, seed=11).items()}, str(tmp_path / "instruct_blip_tiny.safetensors"))
    words = [f"[unused{i}]" for i in range(bcfg.qf.vocab)]
    words[0], words[100], words[101], words[102] = "[PAD]", "[UNK]", "[CLS]", "[SEP]"
    for i, w in enumerate(("is", "there", "a", "thing", "in", "the", "image", "?", "please", "answer", "this", "question", "with", "one", "word.")):
        words[200 + i] = w
    t = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="[UNK]"))
    t.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    t.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", special_tokens=[("[CLS]", 101), ("[SEP]", 102)])
    bert_dir = tmp_path / "bert-tiny"
    PreTrainedTokenizerFast(tokenizer_object=t, unk_token="[UNK]", pad_token="[PAD]", cls_token="[CLS]", sep_token="[SEP]").save_pretrained(str(bert_dir))
    qfile = tmp_path / "pope.json"
    with open(qfile, "w") as f:
        for q in range(10):
            f.write(json.dumps({"question_id": q, "image": f"im{q % 5}.png", "text": f"is there a w{q} in the image ?", "label": ("yes", "no")[q % 2]}) + "\n")
    out = str(tmp_path / "out" / "blip.jsonl")
    p = subprocess.run([sys.executable, "-m", "llava_align_amd.blip_driver", "--blip-checkpoint", str(tmp_path / "instruct_blip_tiny.safetensors"), "--llm-path",
                        info["ckpt"], "--bert-tokenizer", str(bert_dir), "--image-folder", info["images"], "--question-file", str(qfile), "--answers-file", out,
                        "--use_cd", "--noise_step", "500", "--cd_beta", "0.1", "--seed", "7", "--batch", "4"], capture_output=True, text=True, env=_env(),
                       timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    rep = json.loads(p.stdout[p.stdout.index("{"):])
    assert rep["n_answers"] == 10 and set(rep["scores"]) == {"string_match", "naive", "noise", "zeros"}
    recs = [json.loads(l) for l in open(out)]
    assert [r["question_id"] for r in recs] == list(range(10))
    assert tuple(recs[0].keys()) == ("question_id", "prompt", "text", "model_id", "image", "naive", "noise", "zeros", "metadata")     # blip_calibrate.py:100-109
    assert recs[0]["prompt"].endswith(" Please answer this question with one word.") and recs[0]["model_id"] == "instruct_blip"


@pytest.mark.gpu
def test_sampling_driver_from_a_checkpoint_directory_alone_and_under_torchrun(tmp_path):
    """`python -m llava_align_amd.sampling_driver` (llava_sampling.py:128-195's arguments; BASELINE config #3's driver): open-ended answers through
    generate_list, 'setting' -> 'default' in the answers file name, the reference's JSONL fields; two ranks write what one rank writes
    (cd_greedy is not a CLI flag there, so the seeded sampled run is compared per rank count with itself: file shape, ids, order)."""
    info = write_checkpoint(str(tmp_path))
    common = ["--model-path", info["ckpt"], "--question-file", info["questions"], "--image-folder", info["images"], "--use_dd", "--use_dd_unk", "--cd_alpha", "1",
              "--cd_beta", "0.1", "--max_new_tokens", "12", "--in-flight", "4", "--no-sweep"]
    one = str(tmp_path / "out" / "one-setting.jsonl")
    p1 = subprocess.run([sys.executable, "-m", "llava_align_amd.sampling_driver", *common, "--answers-file", one], capture_output=True, text=True, env=_env(),
                        timeout=900, cwd=ROOT)
    assert p1.returncode == 0, p1.stderr[-3000:]
    rep = json.loads(p1.stdout.strip().splitlines()[-1])
    assert rep["run"] == "default" and rep["n_answers"] == info["n_questions"] and rep["answers_file"].endswith("one-default.jsonl")
    assert rep["stats"]["in_flight"] == 4 and rep["stats"]["admissions"] >= 2                       # the list went through refilled slots
    a = [json.loads(l) for l in open(rep["answers_file"])]
    assert [x["question_id"] for x in a] == list(range(100, 100 + info["n_questions"]))
    assert tuple(a[0].keys()) == ("question_id", "prompt", "text", "model_id", "image", "metadata") and a[0]["model_id"] == "tiny-llava"   # llava_sampling.py:119-124
    two = str(tmp_path / "out" / "two-setting.jsonl")
    p2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                         "29563", "-m", "llava_align_amd.sampling_driver", *common, "--answers-file", two], capture_output=True, text=True, env=_env(),
                        timeout=900, cwd=ROOT)
    assert p2.returncode == 0, p2.stderr[-3000:]
    b = [json.loads(l) for l in open(two.replace("setting", "default"))]
    assert [x["question_id"] for x in b] == [x["question_id"] for x in a] and all(set(x) == set(a[0]) for x in b)
    # llava_naive.py's call shape: ONE run, the file name as given, greedy at --temperature 0 (plain arg-max: greedy_search is not patched)
    naive = [str(tmp_path / "out" / f"naive{i}.jsonl") for i in range(2)]
    runs = [subprocess.run([sys.executable, "-m", "llava_align_amd.sampling_driver", *common[:6], "--max_new_tokens", "6", "--naive", "--temperature", "0",
                            "--answers-file", f], capture_output=True, text=True, env=_env(), timeout=900, cwd=ROOT) for f in naive]
    assert all(r.returncode == 0 for r in runs), runs[0].stderr[-3000:]
    assert json.loads(runs[0].stdout.strip().splitlines()[-1])["run"] == "naive"
    c, c2 = ([json.loads(l) for l in open(f)] for f in naive)
    assert len(c) == info["n_questions"] and tuple(c[0].keys()) == tuple(a[0].keys()) and [x["text"] for x in c] == [x["text"] for x in c2]
    # --use_cd: the VCD branch (images_cd = add_diffusion_noise(image, noise_step)) through generate_list; only the 'default' run (llava_sampling.py:162-163)
    vcd = str(tmp_path / "out" / "vcd-setting.jsonl")
    p3 = subprocess.run([sys.executable, "-m", "llava_align_amd.sampling_driver", *common[:6], "--use_cd", "--noise_step", "500", "--max_new_tokens", "8",
                         "--in-flight", "4", "--answers-file", vcd], capture_output=True, text=True, env=_env(), timeout=900, cwd=ROOT)
    assert p3.returncode == 0, p3.stderr[-3000:]
    reps = [json.loads(l) for l in p3.stdout.strip().splitlines() if l.startswith("{")]
    assert [r["run"] for r in reps] == ["default"] and reps[0]["stats"]["n_rows"] == 4 and reps[0]["stats"]["admissions"] >= 2
    assert len(open(vcd.replace("setting", "default")).readlines()) == info["n_questions"]
