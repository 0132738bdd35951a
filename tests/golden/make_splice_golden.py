"""Build-container script: runs the REFERENCE's own multimodal splice - `LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal`
and `encode_images` (experiments/llava/model/llava_arch.py:82-204), loaded read-only by path from /root/reference - on seeded inputs
and commits what it returns as tests/golden/splice.npz.  tests/test_splice_golden.py (CPU) checks the splice of tests/ref_llava.py
(the fp32 reference every engine test is compared with) and of tests/hf_llava.py (the HF-module stand-in of the adapter tests) against
these outputs, so the chain  reference function == fixture == ref_llava == engine  has no restated link left.

Only the two builder modules llava_arch.py imports (`multimodal_encoder.builder`, `multimodal_projector.builder`: CLIP / projector
construction, not on the splice path) are stubbed; the vision tower is a stand-in returning seeded patch features, the projector a
seeded nn.Linear - `encode_images` itself is the reference's.  /root/reference does not exist on the GPU box; nothing imports this
module at test time.

    python tests/golden/make_splice_golden.py          # rewrites tests/golden/splice.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("VDD_REFERENCE_ROOT", "/root/reference")
IMAGE_TOKEN_INDEX = -200


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_llava_arch():
    sys.dont_write_bytecode = True
    for name in ("refllava", "refllava.multimodal_encoder", "refllava.multimodal_projector", "llava"):
        pkg = types.ModuleType(name)
        pkg.__path__ = []
        sys.modules[name] = pkg
    enc = types.ModuleType("refllava.multimodal_encoder.builder")
    enc.build_vision_tower = lambda *a, **k: None            # CLIP construction: not on the splice path
    prj = types.ModuleType("refllava.multimodal_projector.builder")
    prj.build_vision_projector = lambda *a, **k: None
    sys.modules[enc.__name__], sys.modules[prj.__name__] = enc, prj
    _load("llava.constants", os.path.join(REF_ROOT, "experiments/llava/constants.py"))
    return _load("refllava.llava_arch", os.path.join(REF_ROOT, "experiments/llava/model/llava_arch.py"))


def inputs(seed=0, vocab=50, d=8, width=6, n_patch=5):
    g = torch.Generator().manual_seed(seed)
    table = torch.randn(vocab, d, generator=g)
    proj_w, proj_b = torch.randn(d, width, generator=g), torch.randn(d, generator=g)
    feats = torch.randn(3, n_patch, width, generator=g)                       # what the tower returns for images 0..2
    ids_main = torch.tensor([[1, 7, 9, IMAGE_TOKEN_INDEX, 21, 22, 23, 3]])
    ids_b2 = torch.tensor([[1, 7, IMAGE_TOKEN_INDEX, 30, 31, 32], [1, 8, 12, IMAGE_TOKEN_INDEX, 40, 41]])
    return dict(table=table, proj_w=proj_w, proj_b=proj_b, feats=feats, ids_main=ids_main, ids_b2=ids_b2)


def main():
    arch = load_llava_arch()
    inp = inputs()

    class Tower(torch.nn.Module):
        """Stand-in for CLIPVisionTower.forward (clip_encoder.py:39-51): `images` [n, 1] carries the index of the seeded feature block."""

        def forward(self, images):
            return inp["feats"][images[:, 0].long()]

    class Inner(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.embed_tokens = torch.nn.Embedding.from_pretrained(inp["table"])
            self.mm_projector = torch.nn.Linear(inp["proj_w"].shape[1], inp["proj_w"].shape[0])
            with torch.no_grad():
                self.mm_projector.weight.copy_(inp["proj_w"]); self.mm_projector.bias.copy_(inp["proj_b"])
            self.tower = Tower()

        def get_vision_tower(self):
            return self.tower

    class Host(arch.LlavaMetaForCausalLM):
        def __init__(self):
            self.model = Inner()
            self.config = types.SimpleNamespace()
            self.device = torch.device("cpu")

        def get_model(self):
            return self.model

    host = Host()
    out = {k: v.numpy() for k, v in inp.items()}
    with torch.no_grad():
        def run(tag, ids, images, past=None, mask=None):
            mask = torch.ones_like(ids) if mask is None else mask
            r_ids, r_mask, r_past, r_emb, r_lab = host.prepare_inputs_labels_for_multimodal(ids, mask, past, None, images)
            out[tag + ".ids_returned"] = np.array(0 if r_ids is None else 1)
            if r_ids is not None:
                out[tag + ".ids"] = r_ids.numpy()
            out[tag + ".mask"] = r_mask.numpy().astype(np.int64)
            out[tag + ".embeds_returned"] = np.array(0 if r_emb is None else 1)
            if r_emb is not None:
                out[tag + ".embeds"] = r_emb.numpy()
            assert r_lab is None and r_past is past
        ids = inp["ids_main"]
        run("main", ids, torch.tensor([[1.0]]))                                           # image 1 spliced in at the -200 slot (:122-163)
        unk = ids.clone(); unk[unk == IMAGE_TOKEN_INDEX] = 0                              # vcd_sample.py:154-155
        run("unk", unk, None)                                                             # images=None: ids as they are (:91-94)
        run("none", ids[ids != IMAGE_TOKEN_INDEX][None], None)                            # vcd_sample.py:160
        past = [(torch.zeros(1, 2, 12, 4), torch.zeros(1, 2, 12, 4))]                     # 12 cached positions
        run("decode", torch.tensor([[5]]), torch.tensor([[1.0]]), past=past, mask=torch.ones(1, 9, dtype=torch.long))   # :92-93
        run("batch2", inp["ids_b2"], torch.tensor([[0.0], [2.0]]))                         # two questions, one image each
        run("no_slot", unk, torch.tensor([[1.0]]))                                        # images given, no -200 in the row (:106-117)
        out["encode_images"] = host.encode_images(torch.tensor([[0.0], [1.0], [2.0]])).numpy()     # :82-85
    np.savez(os.path.join(HERE, "splice.npz"), **out)
    print("wrote", os.path.join(HERE, "splice.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
