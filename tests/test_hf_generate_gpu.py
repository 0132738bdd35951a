"""The install hook under the INSTALLED transformers (5.x): `evolve_vcd_sampling()` then a plain
`model.generate(..., use_dd_unk=..., cd_alpha=..., cd_beta=...)` on a real HF LlamaForCausalLM subclass that follows
the reference's model-side protocol (forward lists the cd kwargs, llava_llama.py:69-79; prepare_inputs_for_generation_cd,
:153-174).  Checked against a cache-free recomputation of every branch + the oracle's per-step arithmetic."""
import pytest
import torch
import transformers
from transformers import LlamaConfig, LlamaForCausalLM

from oracle import vdd_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
IMG = -200


class CdLlama(LlamaForCausalLM):
    """Text-only stand-in for LlavaLlamaForCausalLM: the image slot (-200) is embedded as token 7 when `images` is given."""

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                labels=None, use_cache=None, images=None, images_cd=None, cd_alpha=None, cd_beta=None, use_dd=None,
                use_dd_unk=None, cd_greedy=None, **kw):
        if input_ids is not None:
            input_ids = torch.where(input_ids == IMG, torch.full_like(input_ids, 7 if images is not None else 9), input_ids)
        return super().forward(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                               past_key_values=past_key_values, inputs_embeds=inputs_embeds, use_cache=use_cache, **kw)

    def prepare_inputs_for_generation_cd(self, input_ids, **kw):
        d = self.prepare_inputs_for_generation(input_ids, **kw)
        d["images"] = kw.get("images_cd")                         # llava_llama.py:170
        return d


@pytest.fixture(scope="module")
def model():
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=97, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                      num_key_value_heads=2, max_position_embeddings=256, pad_token_id=0, eos_token_id=None, bos_token_id=1)
    m = CdLlama(cfg).to(DEV).eval()
    for p in m.parameters():                                      # larger logits: clear top-1 margins
        p.data.mul_(3.0)
    return m


@pytest.fixture()
def hooked():
    import llava_align_amd as L
    mixin = transformers.generation.utils.GenerationMixin
    saved = (mixin.__dict__.get("sample"), mixin.__dict__.get("_sample"))
    L.evolve_vcd_sampling()
    yield
    if saved[0] is None:
        del mixin.sample
    else:
        mixin.sample = saved[0]
    mixin._sample = saved[1]


def branch_logits(model, ids, with_image):
    """Last-position logits of a full, cache-free forward."""
    with torch.no_grad():
        return model(input_ids=ids, images=(torch.ones(1) if with_image else None), use_cache=False).logits[:, -1, :]


@pytest.mark.parametrize("mode", ["plain", "dd_unk", "dd", "both", "cd"])
def test_generate_through_installed_transformers(model, hooked, mode):
    ids = torch.tensor([[1, 11, 23, 5, IMG, 40, 41, 77, 12]], device=DEV)
    kw = {"plain": {}, "dd_unk": {"use_dd_unk": True}, "dd": {"use_dd": True}, "both": {"use_dd": True, "use_dd_unk": True},
          "cd": {"images_cd": torch.zeros(1, device=DEV)}}[mode]
    n_new = 5
    out = model.generate(ids, attention_mask=torch.ones_like(ids), images=torch.ones(1, device=DEV), do_sample=True, top_k=1,
                         max_new_tokens=n_new, cd_alpha=1.0, cd_beta=0.1, cd_greedy=True, output_scores=True,
                         return_dict_in_generate=True, **kw)
    seq = out["sequences"]
    assert seq.shape[1] == ids.shape[1] + n_new and torch.equal(seq[:, : ids.shape[1]], ids)
    cur = ids.clone()
    for step in range(n_new):
        v = branch_logits(model, cur, True)
        unk = cur.clone(); unk[unk == IMG] = 0
        none = cur[cur != IMG][None]
        c = d = None
        if mode == "dd_unk":
            c = branch_logits(model, unk, False)
        elif mode == "dd":
            c = branch_logits(model, none, False)
        elif mode == "both":
            c, d = branch_logits(model, unk, False), branch_logits(model, none, False)
        elif mode == "cd":       # images_cd given -> same ids, 'image' present; from step 1 on c == v
            c = branch_logits(model, cur, True)
        want = O.step_scores(v.cpu(), c.cpu() if c is not None else None, d.cpu() if d is not None else None, 1.0, 0.1,
                             O.WarpConfig(top_k=1))
        got = out["scores"][step].cpu()
        fin = torch.isfinite(want[0])
        assert int(fin.sum()) >= 1
        pre = O.step_scores(v.cpu(), c.cpu() if c is not None else None, d.cpu() if d is not None else None, 1.0, 0.1, O.WarpConfig())
        top2 = torch.topk(pre[0], 2).values
        if (top2[0] - top2[1]).item() > 0.05:                     # cache vs cache-free fp32 noise is ~1e-5
            assert torch.equal(torch.isfinite(got[0]), fin)
            assert torch.allclose(got[0][fin], want[0][fin], atol=1e-3)
            assert int(seq[0, ids.shape[1] + step]) == int(torch.softmax(want, -1).argmax())
        cur = torch.cat([cur, seq[:, ids.shape[1] + step: ids.shape[1] + step + 1]], dim=1)


def test_generate_greedy_disables_contrast(model, hooked):
    ids = torch.tensor([[1, 11, 23, IMG, 40]], device=DEV)
    with pytest.warns(UserWarning, match="WITHOUT contrastive"):
        out = model.generate(ids, attention_mask=torch.ones_like(ids), images=torch.ones(1, device=DEV), do_sample=False,
                             max_new_tokens=3, use_dd_unk=True)
    cur = ids.clone()
    for step in range(3):
        tok = branch_logits(model, cur, True).argmax(-1)
        assert int(out[0, ids.shape[1] + step]) == int(tok)
        cur = torch.cat([cur, tok[:, None]], 1)
