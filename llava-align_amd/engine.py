"""Branch-batched LLaVA-1.5 generation engine for MI355X.

What the reference does per question (experiments/eval/calibrate/llava_calibrate.py:161-177 ->
vcd_utils/vcd_sample.py:93-299): B=1, one full eager forward per branch per token (2 for
use_dd_unk, 3 for both), every forward re-reading all weights, the CLIP tower re-run for every
question although POPE asks 6 questions per image.

What this engine does instead (same decoding semantics, same kwargs):
  * many questions in flight; rows of a step = (question, branch) pairs, so the weights stream
    from HBM once per token for ALL questions and branches;
  * ViT + projector features cached per image;
  * prompt-prefix KV sharing: questions with the same [system prompt + image] prefix (and all
    image-free branches, whose prefix is [system prompt + <unk>]) prefill that prefix once; the
    attention kernels read [prefix slot | own slot];
  * lm_head only on last positions (the reference computes logits for all ~635, :103);
  * the per-step tail is the fused vdd_contrast_sample kernel; token/position/length state
    stays on the device and the decode step is replayed as a HIP graph.

Reference-compat details: <unk> is ONE token (#3), so image-free branches are
575 positions shorter; the VCD branch (images_cd) contributes its own logits at step 0 only and
c == v afterwards (#1); use_dd alone or with use_dd_unk adds the image-token-dropped branch.

Every compute op is a hand-written HIP kernel (ops.py), the GEMMs included (csrc/vdd_gemm.hip above 8 rows,
the weight-streaming GEMV kernels below).  16-bit storage - bf16 (BASELINE config #2) or fp16 (the dtype the reference's drivers load
their checkpoints in, experiments/llava/model/builder.py:40) - with fp32 accumulation: `VddLlavaEngine(dtype=...)`.
"""
from __future__ import annotations

import gc
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import ops
from .sampling import WarpSpec, contrast_sample

IMAGE_TOKEN_INDEX = -200   # experiments/llava/constants.py:8


# ------------------------------------------------------------------ configs
@dataclass
class LMConfig:
    d: int = 4096
    n_layers: int = 32
    n_heads: int = 32
    n_kv_heads: int = 32
    head_dim: int = 128
    ffn: int = 11008
    vocab: int = 32000
    rope_theta: float = 10000.0
    eps: float = 1e-5
    max_pos: int = 4096
    qkv_bias: bool = False          # Qwen-7B: c_attn carries a bias (modeling_qwen.py:224-226); Llama/Vicuna: none


@dataclass
class VisionConfig:
    image: int = 336
    patch: int = 14
    width: int = 1024
    layers: int = 24
    select_layer: int = -2          # clip_encoder.py:29-37: hidden_states[-2], CLS dropped
    heads: int = 16
    mlp: int = 4096
    eps: float = 1e-5

    @property
    def n_patches(self):
        return (self.image // self.patch) ** 2

    @property
    def run_layers(self):           # hidden_states[-2] is the output of layer `layers - 1` (1-based), i.e. skip the last
        return self.layers + 1 + self.select_layer


@dataclass
class LlavaConfig:
    lm: LMConfig = field(default_factory=LMConfig)
    vision: VisionConfig = field(default_factory=VisionConfig)
    name: str = "llava-1.5-7b"


def preset(name: str) -> LlavaConfig:
    if name == "llava-1.5-7b":
        return LlavaConfig(LMConfig(), VisionConfig(), name)
    if name == "llava-1.5-13b":
        return LlavaConfig(LMConfig(d=5120, n_layers=40, n_heads=40, n_kv_heads=40, ffn=13824), VisionConfig(), name)
    if name == "qwen-vl-7b-lm":  # BASELINE config #4, LM side only (32 layers, d=4096, V=151936); the Qwen ViT + resampler
        # are not on the north-star path: image slots arrive through `generate(inputs_embeds=...)`
        return LlavaConfig(LMConfig(vocab=151936, eps=1e-6, qkv_bias=True, max_pos=8192), VisionConfig(), name)
    if name == "tiny-qwen":      # test-sized Qwen-shaped LM: qkv bias, V beyond one LDS row (workspace path of the sampling kernel)
        return LlavaConfig(LMConfig(d=256, n_layers=2, n_heads=2, n_kv_heads=2, ffn=512, vocab=151936, eps=1e-6, qkv_bias=True,
                                    max_pos=512),
                           VisionConfig(image=56, patch=14, width=128, layers=3, heads=2, mlp=256), name)
    if name == "tiny":          # test-sized: same structure, every kernel path exercised
        return LlavaConfig(LMConfig(d=256, n_layers=2, n_heads=2, n_kv_heads=2, ffn=512, vocab=1000, max_pos=512),
                           VisionConfig(image=56, patch=14, width=128, layers=3, heads=2, mlp=256), name)
    raise KeyError(name)


# ------------------------------------------------------------------ weights
class LlavaWeights:
    """Flat container of device tensors of ONE 16-bit dtype (bf16 or fp16).  `random()` draws N(0, 0.02) (no
    checkpoints exist on either box); `from_state_dict()` maps HF LLaVA-1.5 parameter names."""

    def __init__(self, cfg: LlavaConfig, device, dtype=torch.bfloat16):
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype       # (fp32 containers feed the tests' torch references)
        self.t: Dict[str, torch.Tensor] = {}

    @staticmethod
    def random(cfg: LlavaConfig, device, seed: int = 0, std: float = 0.02, lm_head_gain: float = 1.0, dtype=torch.bfloat16) -> "LlavaWeights":
        """lm_head_gain scales the output projection: N(0, 0.02) everywhere gives logits of sigma ~ 1.3 (d = 4096), so flat that
        the plausibility mask of a contrastive step keeps ~1000 tokens; a trained LLaVA answers POPE with a few candidates.
        gain 4 (sigma ~ 5) puts a random model in that regime (benchmarks use it; parity tests keep 1)."""
        w = LlavaWeights(cfg, device, dtype)
        g = torch.Generator(device=device).manual_seed(seed)

        def rnd(*shape, s=std):
            return (torch.randn(*shape, device=device, generator=g, dtype=torch.float32) * s).to(dtype)

        def ones(n):
            return (1.0 + torch.randn(n, device=device, generator=g) * 0.02).to(dtype)
        lm, v = cfg.lm, cfg.vision
        qkv_out = (lm.n_heads + 2 * lm.n_kv_heads) * lm.head_dim
        w.t["embed"] = rnd(lm.vocab, lm.d)
        for i in range(lm.n_layers):
            p = f"l{i}."
            w.t[p + "ln1"], w.t[p + "ln2"] = ones(lm.d), ones(lm.d)
            w.t[p + "wqkv"] = rnd(qkv_out, lm.d)
            if lm.qkv_bias:
                w.t[p + "bqkv_lm"] = rnd(qkv_out, s=0.1)
            w.t[p + "wo"] = rnd(lm.d, lm.n_heads * lm.head_dim)
            w.t[p + "wgu"] = rnd(2 * lm.ffn, lm.d)
            w.t[p + "wd"] = rnd(lm.d, lm.ffn)
        w.t["norm"] = ones(lm.d)
        w.t["lm_head"] = rnd(lm.vocab, lm.d, s=std * lm_head_gain)
        pd = 3 * v.patch * v.patch
        pd_pad = (pd + 127) // 128 * 128                         # K of the patch-embed GEMM: a multiple of its 128-deep unit
        pw = torch.zeros(v.width, pd_pad, dtype=dtype, device=device)
        pw[:, :pd] = rnd(v.width, pd)
        w.t["v.patch"] = pw                                      # conv14x14/stride14 as a [width, 588 -> 640] GEMM
        w.t["v.cls"], w.t["v.pos"] = rnd(v.width), rnd(v.n_patches + 1, v.width)
        w.t["v.pre_ln.w"], w.t["v.pre_ln.b"] = ones(v.width), rnd(v.width)
        for i in range(v.run_layers):
            p = f"v{i}."
            for ln in ("ln1", "ln2"):
                w.t[p + ln + ".w"], w.t[p + ln + ".b"] = ones(v.width), rnd(v.width)
            w.t[p + "wqkv"], w.t[p + "bqkv"] = rnd(3 * v.width, v.width), rnd(3 * v.width)
            w.t[p + "wo"], w.t[p + "bo"] = rnd(v.width, v.width), rnd(v.width)
            w.t[p + "fc1"], w.t[p + "b1"] = rnd(v.mlp, v.width), rnd(v.mlp)
            w.t[p + "fc2"], w.t[p + "b2"] = rnd(v.width, v.mlp), rnd(v.width)
        w.t["mm.w1"], w.t["mm.b1"] = rnd(lm.d, v.width), rnd(lm.d)        # mlp2x_gelu (builder.py:33-46)
        w.t["mm.w2"], w.t["mm.b2"] = rnd(lm.d, lm.d), rnd(lm.d)
        return w

    @staticmethod
    def lm_from_state_dict(cfg: LlavaConfig, sd: Dict[str, torch.Tensor], device, dtype=torch.bfloat16) -> "LlavaWeights":
        """The language model alone (HF Llama names; a `LlamaForCausalLM` directory such as InstructBLIP's Vicuna, blip_driver.main):
        no projector, no vision tower - such an engine takes inputs_embeds or text ids."""
        w = LlavaWeights(cfg, device, dtype)
        lm = cfg.lm

        def get(d, k):
            return d[k].detach().to(device=device, dtype=dtype).contiguous()
        w.t["embed"] = get(sd, "model.embed_tokens.weight")
        for i in range(lm.n_layers):
            p, q = f"l{i}.", f"model.layers.{i}."
            w.t[p + "ln1"], w.t[p + "ln2"] = get(sd, q + "input_layernorm.weight"), get(sd, q + "post_attention_layernorm.weight")
            w.t[p + "wqkv"] = torch.cat([get(sd, q + f"self_attn.{n}_proj.weight") for n in ("q", "k", "v")], 0).contiguous()
            if lm.qkv_bias:
                w.t[p + "bqkv_lm"] = torch.cat([get(sd, q + f"self_attn.{n}_proj.bias") for n in ("q", "k", "v")], 0).contiguous()
            w.t[p + "wo"] = get(sd, q + "self_attn.o_proj.weight")
            w.t[p + "wgu"] = torch.cat([get(sd, q + "mlp.gate_proj.weight"), get(sd, q + "mlp.up_proj.weight")], 0).contiguous()
            w.t[p + "wd"] = get(sd, q + "mlp.down_proj.weight")
        w.t["norm"] = get(sd, "model.norm.weight")
        w.t["lm_head"] = get(sd, "lm_head.weight") if "lm_head.weight" in sd else w.t["embed"]       # tie_word_embeddings
        return w

    @staticmethod
    def from_state_dict(cfg: LlavaConfig, sd: Dict[str, torch.Tensor], device, vision_sd: Optional[Dict[str, torch.Tensor]] = None,
                        dtype=torch.bfloat16) -> "LlavaWeights":
        """Maps an original-LLaVA-1.5 checkpoint (experiments/llava/model/builder.py:102-141 loads exactly these names):
        `model.embed_tokens`, `model.layers.N.{self_attn.{q,k,v,o}_proj, mlp.{gate,up,down}_proj, input_layernorm,
        post_attention_layernorm}`, `model.norm`, `lm_head`, `model.mm_projector.{0,2}`; the CLIP tower either inside the
        checkpoint (`model.vision_tower.vision_tower.vision_model...`) or as a separate HF CLIP state dict (`vision_sd`,
        keys `vision_model...`).  q/k/v and gate/up are concatenated for the fused projections."""
        w = LlavaWeights.lm_from_state_dict(cfg, sd, device, dtype)
        v = cfg.vision

        def get(d, k):
            return d[k].detach().to(device=device, dtype=dtype).contiguous()
        w.t["mm.w1"], w.t["mm.b1"] = get(sd, "model.mm_projector.0.weight"), get(sd, "model.mm_projector.0.bias")
        w.t["mm.w2"], w.t["mm.b2"] = get(sd, "model.mm_projector.2.weight"), get(sd, "model.mm_projector.2.bias")
        vs = vision_sd if vision_sd is not None else sd
        # classic checkpoints: `vision_model.`; inside a LLaVA checkpoint: `model.vision_tower.vision_tower.vision_model.`;
        # transformers 5.x CLIPVisionModel.state_dict(): no prefix at all
        vp = next((c for c in ("vision_model.", "model.vision_tower.vision_tower.vision_model.", "model.vision_tower.vision_tower.", "")
                   if c + "embeddings.patch_embedding.weight" in vs), None)
        if vp is None:
            raise KeyError("no CLIP vision tower found in the state dict (looked for *embeddings.patch_embedding.weight)")
        pw = get(vs, vp + "embeddings.patch_embedding.weight").reshape(v.width, -1)        # [width, 3, P, P] -> [width, 3*P*P]
        pd_pad = (pw.shape[1] + 127) // 128 * 128
        w.t["v.patch"] = torch.nn.functional.pad(pw, (0, pd_pad - pw.shape[1])).contiguous()
        w.t["v.cls"] = get(vs, vp + "embeddings.class_embedding")
        w.t["v.pos"] = get(vs, vp + "embeddings.position_embedding.weight")
        w.t["v.pre_ln.w"], w.t["v.pre_ln.b"] = get(vs, vp + "pre_layrnorm.weight"), get(vs, vp + "pre_layrnorm.bias")   # sic (HF spelling)
        for i in range(v.run_layers):
            p, q = f"v{i}.", vp + f"encoder.layers.{i}."
            for a, b in (("ln1", "layer_norm1"), ("ln2", "layer_norm2")):
                w.t[p + a + ".w"], w.t[p + a + ".b"] = get(vs, q + b + ".weight"), get(vs, q + b + ".bias")
            w.t[p + "wqkv"] = torch.cat([get(vs, q + f"self_attn.{n}_proj.weight") for n in ("q", "k", "v")], 0).contiguous()
            w.t[p + "bqkv"] = torch.cat([get(vs, q + f"self_attn.{n}_proj.bias") for n in ("q", "k", "v")], 0).contiguous()
            w.t[p + "wo"], w.t[p + "bo"] = get(vs, q + "self_attn.out_proj.weight"), get(vs, q + "self_attn.out_proj.bias")
            w.t[p + "fc1"], w.t[p + "b1"] = get(vs, q + "mlp.fc1.weight"), get(vs, q + "mlp.fc1.bias")
            w.t[p + "fc2"], w.t[p + "b2"] = get(vs, q + "mlp.fc2.weight"), get(vs, q + "mlp.fc2.bias")
        return w

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.t.values())

    def lm_stream_bytes(self) -> int:
        """Bytes a decode step must stream: every LM weight except the embedding table (gathered)."""
        return sum(t.numel() * 2 for k, t in self.t.items()
                   if (k.startswith("l") and k[1].isdigit()) or k in ("norm", "lm_head"))


def rope_table(lm: LMConfig, device) -> torch.Tensor:
    inv = 1.0 / (lm.rope_theta ** (torch.arange(0, lm.head_dim, 2, dtype=torch.float32) / lm.head_dim))
    ang = torch.arange(lm.max_pos, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.stack([ang.cos(), ang.sin()], dim=-1).contiguous().to(device)


# ------------------------------------------------------------------ vision tower + projector
class VisionTower:
    """CLIP ViT-L/14-336 forward up to hidden_states[select_layer], CLS dropped, then the
    mlp2x_gelu projector (llava_arch.py:82-85, clip_encoder.py:39-51)."""

    def __init__(self, w: LlavaWeights):
        self.w, self.cfg = w, w.cfg.vision
        v = self.cfg
        self.T = v.n_patches + 1
        dh = v.width // v.heads
        if dh != 64:
            raise ValueError(f"vision tower head_dim {dh}: the ViT attention kernel is instantiated for head_dim 64 (CLIP ViT-L/14)")
        self._kv = None
        self.use_graph = False               # set by the engine; a full GRAPH_BATCH of images then replays a HIP graph
        self._graphs: Dict[int, tuple] = {}
        self._seq_desc: Dict[int, torch.Tensor] = {}

    def _seqs(self, n):
        if n not in self._seq_desc:
            T = self.T
            self._seq_desc[n] = torch.tensor([[i * T, T, 0, i, 0, 0] for i in range(n)], dtype=torch.int32, device=self.w.device)
        return self._seq_desc[n]

    def _kv_cache(self, n_img):
        v = self.cfg
        if self._kv is None or self._kv[0].shape[0] < n_img:
            n_alloc = max(n_img, self.GRAPH_BATCH)
            mk = lambda: torch.empty(n_alloc, v.heads, self.T, v.width // v.heads, dtype=self.w.dtype, device=self.w.device)
            self._kv = (mk(), mk())
            self._graphs.clear()                 # a captured forward points at the old buffers
        return self._kv

    GRAPH_BATCH = 16
    GRAPH_SIZES = (1, 16)                    # batch sizes replayed from a captured graph: one question's image, a full batch

    @torch.no_grad()
    def __call__(self, images: torch.Tensor) -> torch.Tensor:
        """images [n, 3, S, S] (any float dtype) -> projected patch features [n, n_patches, d_lm] in the model dtype.
        Batches of GRAPH_SIZES images replay a captured HIP graph: the tower is ~400 small launches per batch and runs
        when nothing else is queued, so issued from Python it is launch-bound (98 ms of host time for 43 ms of GPU work
        per 64 images, tools/host_phase_probe.py)."""
        n = images.shape[0]
        if self.use_graph and n in self.GRAPH_SIZES and self.w.device.type == "cuda":
            st = self._graphs.get(n)
            if st is None:
                self._kv_cache(max(n, self.GRAPH_BATCH))
                g_in = torch.empty(n, *images.shape[1:], dtype=self.w.dtype, device=self.w.device)
                g_in.copy_(images)
                side = torch.cuda.Stream(self.w.device)
                side.wait_stream(torch.cuda.current_stream(self.w.device))
                with torch.cuda.stream(side):
                    for _ in range(2):
                        self._forward(g_in)
                torch.cuda.current_stream(self.w.device).wait_stream(side)
                g = torch.cuda.CUDAGraph()
                gc.collect()
                gc_on = gc.isenabled()
                gc.disable()
                try:
                    with torch.cuda.graph(g, stream=side):          # the warm-up stream: its GEMM workspace exists (ops._gemm_workspace)
                        g_out = self._forward(g_in)
                finally:
                    if gc_on:
                        gc.enable()
                st = self._graphs[n] = (g, g_in, g_out)
            g, g_in, g_out = st
            g_in.copy_(images.to(self.w.device, non_blocking=True))      # H2D in the caller's dtype, cast to the model dtype on the device
            g.replay()
            return g_out.clone()
        return self._forward(images)

    @torch.no_grad()
    def _forward(self, images: torch.Tensor) -> torch.Tensor:
        """Every op is a kernel of this package: im2col -> patch-embed GEMM -> class token + positions -> pre-LN -> layers
        (LN, qkv GEMM + bias, split into the attention's q / K / V layout, flash attention, o-proj GEMM + bias + residual, LN,
        fc1 GEMM + bias + quick-GELU, fc2 GEMM + bias + residual) -> projector GEMMs (+ bias + GELU, + bias)."""
        v, t = self.cfg, self.w.t
        n = images.shape[0]
        dev = self.w.device
        x = images.to(device=dev)
        if x.dtype not in (torch.float32, torch.float16, torch.bfloat16):
            x = x.float()
        T, H, D = self.T, v.heads, v.width // v.heads
        patches = ops.vit_im2col(x.contiguous(), v.patch, t["v.patch"].shape[1], dtype=self.w.dtype)          # [n * 576, 640]
        emb = ops.gemm(patches, t["v.patch"])                                              # CLIP's patch conv has no bias
        h = ops.vit_assemble(emb, t["v.cls"], t["v.pos"], n, T)
        h = ops.layernorm(h, t["v.pre_ln.w"], t["v.pre_ln.b"], v.eps)
        kc, vc = self._kv_cache(n)
        seqs = self._seqs(n)
        for i in range(v.run_layers):
            p = f"v{i}."
            a = ops.layernorm(h, t[p + "ln1.w"], t[p + "ln1.b"], v.eps)
            qkv = ops.gemm(a, t[p + "wqkv"], bias=t[p + "bqkv"], epi=ops.EPI_BIAS)
            q = ops.vit_qkv_split(qkv, kc, vc, n, T, H, D)
            att = ops.flash_attention(q, kc, vc, seqs, n, T, H, H, D, causal=False)
            h = ops.gemm(att, t[p + "wo"], bias=t[p + "bo"], resid=h, epi=ops.EPI_BIAS_RESID)
            a = ops.layernorm(h, t[p + "ln2.w"], t[p + "ln2.b"], v.eps)
            f = ops.gemm(a, t[p + "fc1"], bias=t[p + "b1"], epi=ops.EPI_BIAS_QUICK_GELU)
            h = ops.gemm(f, t[p + "fc2"], bias=t[p + "b2"], resid=h, epi=ops.EPI_BIAS_RESID)
        # the projector runs on all T rows (the class row is 1 of 577) and the class row is dropped by a VIEW ('patch', clip_encoder.py:33-37)
        z = ops.gemm(h, t["mm.w1"], bias=t["mm.b1"], epi=ops.EPI_BIAS_GELU)
        z = ops.gemm(z, t["mm.w2"], bias=t["mm.b2"], epi=ops.EPI_BIAS)
        return z.view(n, T, -1)[:, 1:]


# ------------------------------------------------------------------ language model
class KVCache:
    """Pools per layer, all in the model dtype:
    `own`   one COMPACT slot per (question, branch), [n_own, n_kv_heads, t_own, head_dim]: token t at index t - prefix_len, so a slot only
            holds the question's own ~25 prompt tokens + the generated ones instead of a full-length context;
    shared prompt prefixes (system prompt + image patches / + <unk>), in ONE of two forms:
    `frag`  (frag_only=True: the decode steps attend the prefixes through the grouped MFMA pass) a fragment-major image per layer,
            [n_pre, n_kv_heads, 2 t_pre, head_dim]: one 32-KiB block of MFMA operand images per 64-key chunk (vdd_prefix_fragments).
            The row-major K / V a prefill layer writes and its suffix pass reads live in ONE scratch pair shared by all layers
            (`kp[i]` / `vp[i]` are the same tensor for every i): the prefill walks prefix and suffix rows layer by layer together
            and converts the scratch into `pfrag[i]` before the next layer overwrites it.  Round 3 kept both forms for every layer:
            129 slots x 4 x 168 MB = 87 GB of the 189 GB peak of a 768-question batch, half of it read once;
    `pre`   (frag_only=False: few rows in flight / nothing worth grouping - the decode kernels read row-major prefixes) per-layer
            row-major pools [n_pre, n_kv_heads, t_pre, head_dim], token t at index t, and no fragment image at all."""

    def __init__(self, lm: LMConfig, n_pre: int, t_pre: int, n_own: int, t_own: int, device, dtype=torch.bfloat16, frag_only: bool = False):
        self.dtype, self.frag_only = dtype, frag_only
        self.n_pre, self.t_pre, self.n_own, self.t_own = n_pre, t_pre, n_own, t_own
        mk1 = lambda n, t: torch.empty((max(n, 1), lm.n_kv_heads, t, lm.head_dim), dtype=dtype, device=device)
        mk = lambda n, t: [mk1(n, t) for _ in range(lm.n_layers)]
        if frag_only:
            k1, v1 = mk1(n_pre, t_pre), mk1(n_pre, t_pre)
            self.kp, self.vp = [k1] * lm.n_layers, [v1] * lm.n_layers
            self.pfrag = mk(n_pre, 2 * t_pre)
        else:
            self.kp, self.vp = mk(n_pre, t_pre), mk(n_pre, t_pre)
            self.pfrag = [None] * lm.n_layers
        self.ko, self.vo = mk(n_own, t_own), mk(n_own, t_own)

    @staticmethod
    def bytes_needed(lm: LMConfig, n_pre, t_pre, n_own, t_own, dtype, frag_only) -> int:
        per = lm.n_kv_heads * lm.head_dim * (2 if dtype in (torch.bfloat16, torch.float16) else 4)
        pre = max(n_pre, 1) * t_pre * per * 2
        return lm.n_layers * max(n_own, 1) * t_own * per * 2 + (lm.n_layers * pre + pre if frag_only else lm.n_layers * pre)

    def fits(self, n_pre, t_pre, n_own, t_own, frag_only):
        return (frag_only == self.frag_only and n_pre <= self.n_pre and t_pre <= self.t_pre and n_own <= self.n_own and t_own <= self.t_own)

    def repack_own(self, keep_slots: torch.Tensor, live_len: int, t_own: int):
        """Row retirement / growth of the own pools: keep the slots `keep_slots` (int64 device tensor, in their new order: old slot
        keep_slots[i] becomes slot i), with room for `t_own` own tokens each; the first `live_len` rows of every kept slot are
        copied.  One layer at a time (the old tensor of a layer is released before the next layer's new one is allocated), so the
        peak is the new pools + one layer of the old ones.  Tensors change address: captured steps over this cache are stale."""
        t_own = (max(t_own, 64) + 15) // 16 * 16
        n = int(keep_slots.numel())
        live = min(live_len, self.t_own, t_own)
        for pool in (self.ko, self.vo):
            for i in range(len(pool)):
                old = pool[i]
                new = torch.empty((max(n, 1), old.shape[1], t_own, old.shape[3]), dtype=old.dtype, device=old.device)
                if n and live:
                    new[:n, :, :live].copy_(old[:, :, :live].index_select(0, keep_slots))
                pool[i] = new
                del old
        self.n_own, self.t_own = max(n, 1), t_own

    def nbytes(self):
        seen, total = set(), 0
        for pool in (self.kp, self.vp, self.pfrag, self.ko, self.vo):
            for t in pool:
                if t is not None and t.data_ptr() not in seen:
                    seen.add(t.data_ptr())
                    total += t.numel() * t.element_size()
        return total


def h2d_int32(device, *arrays):
    """Several small host integer arrays -> device int32 tensors through ONE pinned staging buffer (torch's caching host
    allocator recycles it) and ONE async copy.  Each `torch.tensor(list, device=...)` is a synchronous pageable copy; a
    dozen of them sat in every generate()."""
    flat = [torch.as_tensor(a, dtype=torch.int32).reshape(-1) for a in arrays]
    shapes = [tuple(torch.as_tensor(a).shape) if not torch.is_tensor(a) else tuple(a.shape) for a in arrays]
    sizes = [int(f.numel()) for f in flat]
    host = torch.empty(sum(sizes), dtype=torch.int32, pin_memory=True)
    if flat:
        torch.cat(flat, out=host)
    dev_buf = host.to(device, non_blocking=True)
    out, o = [], 0
    for shape, n in zip(shapes, sizes):
        out.append(dev_buf[o:o + n].view(*shape) if n else dev_buf[o:o])
        o += n
    return out


def h2d_long(device, values) -> torch.Tensor:
    """A host list of indices -> an int64 device tensor without blocking the host: `torch.tensor(list, device=...)` is a pageable copy that
    WAITS for everything queued on the stream - inside generate() that was the vision tower (57 ms of a 768-question POPE call during which
    the host planned nothing, and the GPU then idled until the prefill's first launch: tools/call_gaps.py)."""
    return h2d_int32(device, values)[0].long()


def prompt_plus_answer(device, ids_list, gen: torch.Tensor) -> List[torch.Tensor]:
    """The `sequences` of a call: per question [prompt ids | generated row] (vcd_sample.py:263 appends to input_ids; prompts given as
    embeddings have no ids: the new tokens alone, as HF returns them).  ONE upload + ONE gather into a flat buffer whose slices are the
    rows - a torch.cat per question was 768 launches (9 ms of host time behind the last decode step of the bench batch: tools/call_gaps.py)."""
    Q, T = int(gen.shape[0]), int(gen.shape[1])
    lens = [len(r) for r in ids_list]
    P = sum(lens)
    if P == 0:
        return [gen[q] for q in range(Q)]
    idx = np.empty(P + Q * T, dtype=np.int32)
    o = p0 = 0
    ar_t = np.arange(T, dtype=np.int32)
    for q, n in enumerate(lens):
        idx[o:o + n] = np.arange(p0, p0 + n, dtype=np.int32)
        idx[o + n:o + n + T] = ar_t + (P + q * T)
        o += n + T
        p0 += n
    flat, idx_d = h2d_int32(device, np.fromiter((t for r in ids_list for t in r), dtype=np.int32, count=P), idx)
    out = torch.cat([flat.long(), gen.reshape(-1)])[idx_d.long()]
    return list(out.split([n + T for n in lens]))


def grouping_pays(groups, rows, min_saved=0.25) -> bool:
    """The grouped prefix pass reads a shared prefix once per group instead of once per row; it only beats the per-row
    split-KV kernel when that removes a real share of the step's KV bytes.  BASELINE config #3 (one image per question:
    the only shared prefixes are the 35-token system prompts of the image-free branches) saves 7 % and runs 13 % FASTER
    ungrouped (tools/config3_probe.py: 2,212 vs 2,511 tokens/s); the POPE batch (6 questions per image) saves 72 %."""
    saved = sum((n_rows - 1) * plen for _, n_rows, _, plen in groups)
    total = sum(r[1] for r in rows)
    return total > 0 and saved >= min_saved * total


class LanguageModel:
    def __init__(self, w: LlavaWeights):
        self.w, self.cfg = w, w.cfg.lm
        self.cs = rope_table(self.cfg, w.device)
        # A vocabulary that is not a multiple of 8 rows (resize_token_embeddings(len(tokenizer)) after add_tokens, builder.py:127-132:
        # 32001 ... 32003): the output projection runs on a zero-padded copy of lm_head (the GEMM writes whole 8-byte quads) and the
        # logits are a [rows, V] VIEW of its [rows, V_pad] result - the sampling kernel takes any row stride and never sees the padding.
        V = self.cfg.vocab
        self.lm_head = w.t["lm_head"]
        if V % 8 != 0:
            padded = torch.zeros((V + 7) // 8 * 8, self.lm_head.shape[1], dtype=self.lm_head.dtype, device=self.lm_head.device)
            padded[:V] = self.lm_head
            self.lm_head = padded

    def _head(self, a=None, resid=None, ss=None):
        """Final norm (fused for a few rows) + output projection -> [rows, V] logits."""
        c = self.cfg
        y = ops.linear(a, self.lm_head) if a is not None else ops.linear_normed(resid, ss, self.w.t["norm"], c.eps, self.lm_head)
        return y if y.shape[1] == c.vocab else y[:, :c.vocab]

    @torch.no_grad()
    def prefill(self, passes: List[dict], kv: KVCache, frag_plen: Optional[torch.Tensor] = None):
        """The prompt prefill.  passes: one dict per set of packed sequences, walked LAYER BY LAYER together, in order - the prefix pass
        (`to_prefix_pool=True`: the shared prompt prefixes, K/V into the prefix pool) before the suffix pass whose sequences continue
        them - each with x [T, d] packed embeddings; pos (rotary) / cpos (index inside the slot) / slot int32 [T]; seqs [n_seq, 6]
        (ops.flash_attention); n_seq, max_tq; optionally last_rows int64 [n_seq] + last_seqs [n_seq, 6] (one-query descriptors) and
        packs (ops.flash_packs).  Returns per pass (residual, delta) of the LAST token of every sequence - the final hidden state is
        their sum (added inside the last norm) - or (None, None) for a pass without last_rows.  The last decoder layer only computes
        what someone reads: its K/V for every token (decode attends them), but attention / o-proj / MLP for the last token of each
        sequence alone, and nothing past the KV write in a pass whose hidden states feed no logits (the reference runs all
        positions through everything, llava_llama.py:88-103).
        frag_plen (int32 [n_prefix slots], with kv.frag_only): after a layer's passes its row-major prefix K/V - the scratch every
        layer shares - is converted into that layer's fragment image for the decode steps."""
        c, t = self.cfg, self.w.t
        H, Hkv, D = c.n_heads, c.n_kv_heads, c.head_dim
        state = [dict(resid=p["x"], delta=None, done=False) for p in passes]
        for i in range(c.n_layers):
            pfx = f"l{i}."
            final = i == c.n_layers - 1
            for p, st in zip(passes, state):
                if st["done"]:
                    continue
                resid, delta = st["resid"], st["delta"]
                new_resid = torch.empty_like(resid) if delta is not None else None
                a = ops.rmsnorm(resid, t[pfx + "ln1"], c.eps, delta=delta, resid_out=new_resid)
                resid = new_resid if new_resid is not None else resid
                qkv = ops.linear(a, t[pfx + "wqkv"], bias=t[pfx + "bqkv_lm"] if c.qkv_bias else None)
                kw_, vw_ = (kv.kp[i], kv.vp[i]) if p["to_prefix_pool"] else (kv.ko[i], kv.vo[i])
                q = ops.rope_kv_write(qkv, p["pos"], p["slot"], self.cs, kw_, vw_, H, Hkv, D, cpos=p["cpos"])
                if p.get("parent_copy") is not None:          # two-level prefixes: the rows in front of an image prefix = its system prompt's
                    dst, src, n = p["parent_copy"]             # (this layer's K / V of the parent slot, computed by the pass before)
                    kw_[dst, :, :n] = kw_[src, :, :n]
                    vw_[dst, :, :n] = vw_[src, :, :n]
                if final and p.get("keep_q"):
                    p["q_last"] = q                                # the rotated queries of the last layer, every row (attention maps on request)
                last_rows = p.get("last_rows")
                if final and last_rows is None:
                    st.update(resid=None, delta=None, done=True)       # only this layer's K/V were still needed
                    continue
                if final:
                    q, resid = q[last_rows].contiguous(), resid[last_rows].contiguous()
                sq, tq = (p["last_seqs"], 1) if final else (p["seqs"], p["max_tq"])
                if p.get("packs") is not None:
                    att = ops.flash_attention_packed(q, kw_, vw_, sq, p["packs"], p["packs"].shape[0], H, Hkv, D, k_prefix=kv.kp[i], v_prefix=kv.vp[i])
                else:
                    att = ops.flash_attention(q, kw_, vw_, sq, p["n_seq"], tq, H, Hkv, D, causal=True, k_prefix=kv.kp[i], v_prefix=kv.vp[i],
                                              own_row_offset=p.get("own_row_offset", 0))
                o = ops.linear(att, t[pfx + "wo"])
                new_resid = torch.empty_like(resid)
                a = ops.rmsnorm(resid, t[pfx + "ln2"], c.eps, delta=o, resid_out=new_resid)
                st["resid"] = new_resid
                st["delta"] = ops.linear(ops.swiglu_linear(a, t[pfx + "wgu"]), t[pfx + "wd"])
            if frag_plen is not None:
                ops.prefix_fragments(kv.kp[i], kv.vp[i], kv.pfrag[i], frag_plen)
        return [(st["resid"], st["delta"]) for st in state]

    @torch.no_grad()
    def logits(self, resid, delta, rows: Optional[torch.Tensor] = None) -> torch.Tensor:
        """final norm + lm_head on the selected rows only (the reference computes all positions, llava_llama.py:103)."""
        c, t = self.cfg, self.w.t
        if rows is not None:
            resid, delta = resid[rows].contiguous(), delta[rows].contiguous()
        a = ops.rmsnorm(resid, t["norm"], c.eps, delta=delta)
        return self._head(a)

    fuse_norms = True         # few rows in flight: the RMSNorm launches ride inside the projections around them
    @torch.no_grad()
    def _decode_step_few_rows(self, resid, pos, cpos, slot, attn_rows, kv):
        """One question (2-3 branch rows) up to 16 rows: 5 launches per layer instead of 7.  The attention-output and MLP-down
        projections write the residual stream themselves (+ per-block sums of squares of its rows), and the projections that read a
        normalised input (qkv, gate/up, lm_head) normalise it as it loads (ops.linear_resid_ss / linear_normed): the two 6.8-us
        RMSNorm launches of a layer - a tenth of a one-question step - are gone; only layer 0 normalises the embeddings with the
        stand-alone kernel."""
        c, t = self.cfg, self.w.t
        H, Hkv, D = c.n_heads, c.n_kv_heads, c.head_dim
        ss = None
        for i in range(c.n_layers):
            p = f"l{i}."
            bias = t[p + "bqkv_lm"] if c.qkv_bias else None
            if ss is None:
                qkv = ops.linear(ops.rmsnorm(resid, t[p + "ln1"], c.eps), t[p + "wqkv"], bias=bias)
            else:
                qkv = ops.linear_normed(resid, ss, t[p + "ln1"], c.eps, t[p + "wqkv"], bias=bias)
            att = ops.decode_attention_fused(qkv, pos, cpos, slot, self.cs, kv.ko[i], kv.vo[i], attn_rows, H, Hkv, D,
                                             k_prefix=kv.kp[i], v_prefix=kv.vp[i])
            resid, ss = ops.linear_resid_ss(att, t[p + "wo"], resid)
            act = ops.swiglu_linear_normed(resid, ss, t[p + "ln2"], c.eps, t[p + "wgu"])
            resid, ss = ops.linear_resid_ss(act, t[p + "wd"], resid)
        return self._head(resid=resid, ss=ss)

    @torch.no_grad()
    def decode_step(self, tokens: torch.Tensor, pos: torch.Tensor, cpos: torch.Tensor, slot: torch.Tensor, attn_rows: torch.Tensor,
                    kv: KVCache, grouping: Optional[dict] = None, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One token for each of M rows: tokens int64 [M], pos/slot int32 [M], attn_rows int32 [M,4] (slot, len, pslot, plen)
        with len already counting the new token.  Returns logits [M, V].  `workspace`: split-KV partials buffer of the
        ungrouped attention; a captured step must own it (the module-level one is re-allocated when a later call needs more)."""
        c, t = self.cfg, self.w.t
        H, Hkv, D = c.n_heads, c.n_kv_heads, c.head_dim
        if kv.frag_only and grouping is None:
            # a fragment-only cache keeps ONE row-major prefix scratch for all layers (it holds the last layer's K / V after the prefill):
            # only the grouped pass, which reads the per-layer fragment images, may decode from it
            raise ValueError("decode_step: a frag_only KVCache decodes through the grouped attention only (pass `grouping`)")
        resid = ops.embed(tokens, t["embed"])
        M = tokens.shape[0]
        if self.fuse_norms and grouping is None and M <= ops.FUSED_ATTN_MAX_M and D == 128 and c.ffn % 128 == 0 and c.n_layers > 0:
            # few rows: the norm-fused five-launch layer or the seven-launch layer, whichever makes THIS step faster - measured once per
            # (rows, width, dtype) on the real step (both forms write the same K / V and leave the runner's state alone), then persisted
            # (the seven-launch form updates the residual stream in place: each timed run gets its own copy)
            forms = lambda: dict(fused=lambda i: self._decode_step_few_rows(resid.clone(), pos, cpos, slot, attn_rows, kv),
                                 plain=lambda i: self._decode_layers(resid.clone(), pos, cpos, slot, attn_rows, kv, None, workspace),
                                 device=self.w.device, n_rot=1, iters=3)
            if ops.norm_fused_pays(M, c.d, self.w.dtype, forms):
                return self._decode_step_few_rows(resid, pos, cpos, slot, attn_rows, kv)
        return self._decode_layers(resid, pos, cpos, slot, attn_rows, kv, grouping, workspace)

    @torch.no_grad()
    def _decode_layers(self, resid, pos, cpos, slot, attn_rows, kv, grouping, workspace):
        """The decoder layers + head of a decode step with stand-alone RMSNorm launches (seven launches per layer at a few rows)."""
        c, t = self.cfg, self.w.t
        H, Hkv, D = c.n_heads, c.n_kv_heads, c.head_dim
        M = resid.shape[0]
        delta = None
        for i in range(c.n_layers):
            p = f"l{i}."
            if delta is None:
                a = ops.rmsnorm(resid, t[p + "ln1"], c.eps)
            else:
                a = ops.rmsnorm(resid, t[p + "ln1"], c.eps, delta=delta, resid_out=resid)
            qkv = ops.linear(a, t[p + "wqkv"], bias=t[p + "bqkv_lm"] if c.qkv_bias else None)
            if grouping is None and M <= ops.fused_attention_rows() and D == 128:
                # a few rows (one question in flight): RoPE + KV write + attention + merge in one launch
                att = ops.decode_attention_fused(qkv, pos, cpos, slot, self.cs, kv.ko[i], kv.vo[i], attn_rows, H, Hkv, D,
                                                 k_prefix=kv.kp[i], v_prefix=kv.vp[i])
            elif grouping is not None:    # rows sharing a prompt prefix attend it once per group (MFMA), own tokens per row
                q = ops.rope_kv_write(qkv, pos, slot, self.cs, kv.ko[i], kv.vo[i], H, Hkv, D, cpos=cpos)
                att = ops.decode_attention_grouped(q, kv.ko[i], kv.vo[i], kv.kp[i], kv.vp[i], attn_rows, grouping["groups"],
                                                   grouping["group_rows"], grouping["items"], grouping["n_items"], H, Hkv, D,
                                                   kv.t_pre, kv.t_own, workspace=grouping["workspace"],
                                                   prefix_frag=kv.pfrag[i], chunks_per_item=grouping["cpi"])
            else:
                q = ops.rope_kv_write(qkv, pos, slot, self.cs, kv.ko[i], kv.vo[i], H, Hkv, D, cpos=cpos)
                att = ops.decode_attention(q, kv.ko[i], kv.vo[i], attn_rows, H, Hkv, D, k_prefix=kv.kp[i], v_prefix=kv.vp[i],
                                           max_len=kv.t_pre + kv.t_own, workspace=workspace)
            o = ops.linear_to_norm(att, t[p + "wo"])
            a = ops.rmsnorm(resid, t[p + "ln2"], c.eps, delta=o, resid_out=resid)
            delta = ops.linear_to_norm(ops.swiglu_linear(a, t[p + "wgu"]), t[p + "wd"])
        a = ops.rmsnorm(resid, t["norm"], c.eps, delta=delta)
        return self._head(a)


# ------------------------------------------------------------------ generation
@dataclass
class GenerateOutput:
    sequences: List[torch.Tensor]            # per question: prompt ids (with -200) + generated ids
    tokens: torch.Tensor                     # [Q, n_new] int64 generated ids (pad after EOS)
    scores: Optional[List[torch.Tensor]]     # per step [Q, V] post-warp scores when output_scores
    top_prob: Optional[torch.Tensor] = None  # step-0 top-10 softmax(scores) (metrics.py:102-104)
    top_tok: Optional[torch.Tensor] = None
    stats: dict = field(default_factory=dict)
    attentions: Optional["StepAttentions"] = None      # output_attentions=True with ONE question: what llava_calibrate.py:180 reads
    branch_top: Optional[Dict[str, tuple]] = None      # branch_priors=True: {"unk" / "none": (top_tok, top_prob)} of that branch's OWN step-0 distribution

    def __getitem__(self, k):
        if k == "attentions" and self.attentions is not None:
            return self.attentions
        if k in ("attentions", "hidden_states"):
            raise KeyError(f"{k}: not produced by the native engine (attention is computed by flash-style kernels that never materialise the "
                           f"maps).  llava_calibrate.py:180-182 reads model_outputs['attentions'][0][-1] and averages it in live code, for a "
                           f"plot whose call (:183) is commented out: delete those lines in a driver ported to the engine, or use the "
                           f"generic evolve_vcd_sampling() path on an HF model for the maps")
        return getattr(self, k)


class StepAttentions:
    """`model_outputs['attentions']` as far as the reference's driver reads it (llava_calibrate.py:180: `['attentions'][0][-1]`): a
    sequence over generation steps whose step 0 is a sequence over layers whose LAST entry is the materialised map [1, H, T, T] of the
    main branch's prompt (ops.attention_probs).  HF returns every layer of every step; the engine's flash-style kernels build none of
    them, and only this one is computed afterwards from the prefill's rotated queries and cached keys - any other index says so."""

    class _Layers:
        def __init__(self, last, n_layers):
            self.last, self.n = last, n_layers

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            if i in (-1, self.n - 1):
                return self.last
            raise IndexError(f"attentions[0][{i}]: only the LAST layer's map of step 0 is materialised (what llava_calibrate.py:180 reads); the "
                             f"other layers are computed by flash-style kernels that never build the matrix")

    def __init__(self, last_layer_map, n_layers, n_steps):
        self.step0, self.n_steps = StepAttentions._Layers(last_layer_map, n_layers), n_steps

    def __len__(self):
        return self.n_steps

    def __getitem__(self, i):
        if i == 0 or (i < 0 and i == -self.n_steps):
            return self.step0
        raise IndexError(f"attentions[{i}]: only step 0 is materialised (llava_calibrate.py:180 reads ['attentions'][0][-1])")


def group_rows_by_prefix(rows):
    """rows: [[slot, len, pslot, plen], ...] -> (groups [[row_off, n_rows, pslot, plen]], flat member row ids);
    rows without a prefix belong to no group."""
    by = {}
    for i, (_, _, ps, pl) in enumerate(rows):
        if pl > 0:
            by.setdefault((ps, pl), []).append(i)
    groups, members = [], []
    for (ps, pl), idx in by.items():
        groups.append([len(members), len(idx), ps, pl])
        members += idx
    return groups, members


class _DecodeRunner:
    """One decode step over static device buffers, so that it can be captured once and replayed as a HIP graph:
    token broadcast -> 32 x (rmsnorm, qkv, rope+KV write, attention, o-proj, rmsnorm, gate/up, silu*mul, down)
    -> final norm -> lm_head -> fused contrastive sampling tail -> state update.  No host interaction."""

    def __init__(self, eng, Q, nb, max_new, tail, kv):
        dev = eng.device
        self.eng, self.Q, self.nb, self.tail, self.kv = eng, Q, nb, tail, kv
        R = nb * Q
        i32 = dict(dtype=torch.int32, device=dev)
        self.tokens_rows = torch.zeros(R, dtype=torch.long, device=dev)
        self.pos, self.cpos, self.slot, self.rows = torch.zeros(R, **i32), torch.zeros(R, **i32), torch.zeros(R, **i32), torch.zeros(R, 4, **i32)
        self.tok = torch.zeros(Q, dtype=torch.long, device=dev)
        self.unfinished = torch.ones(Q, dtype=torch.long, device=dev)
        self.gen = torch.zeros(Q, max_new + 1, dtype=torch.long, device=dev)     # +1: slack column, never returned
        self.step_idx = torch.zeros(1, dtype=torch.long, device=dev)
        self.ctr = torch.zeros(1, dtype=torch.long, device=dev)
        self.status, self.status0, self._st = torch.zeros(Q, **i32), torch.zeros(Q, **i32), torch.zeros(Q, **i32)
        self.scores_buf = torch.empty(Q, eng.cfg.lm.vocab, dtype=eng.dtype, device=dev) if tail["output_scores"] else None
        self.graph = None
        self._tried = False
        self.grouping = None
        # admission mode (VddLlavaEngine.generate_list): question slots are refilled while the others keep decoding, so every slot has its
        # own first step - gen column of slot q at global step s = s - s0[q]; finished slots stay in the batch, frozen (see body)
        self.max_new = max_new
        self.s0 = torch.zeros(Q, dtype=torch.long, device=dev) if tail.get("admit") else None
        # logits-processor stage (vcd_sample.py:197): per-row EOS floor, stop-word flags, repetition-penalty history
        pr = tail.get("proc") or {}
        self.proc = pr
        self.eos_min = torch.zeros(Q, **i32) if pr.get("eos_min") else None
        self.force = torch.zeros(Q, **i32) if pr.get("stop") is not None else None
        self.prompt_tail = torch.full((Q, pr["stop"].max_len), -1, dtype=torch.long, device=dev) if pr.get("stop") is not None else None
        self.prompt_ids = (torch.full((Q, max(1, pr["hist_len"])), -1, dtype=torch.long, device=dev)
                           if (pr.get("rep") is not None or pr.get("python")) else None)
        self.prompt_cols = 0
        lm = eng.cfg.lm
        # the step's own partials buffer (also for the ungrouped split-KV pass): a captured graph must not point into a
        # shared buffer that a later, larger call re-allocates
        self.workspace = ops.attention_workspace(R, lm.n_heads, lm.head_dim, kv.t_pre + (kv.t_own + 63) // 64 * 64, dev)
        if tail.get("n_groups", 0) > 0:
            self.grouping = dict(groups=torch.zeros(max(1, tail["n_groups"]), 4, **i32), group_rows=torch.zeros(R, **i32),
                                 n_groups=tail["n_groups"], items=torch.zeros(max(1, tail["n_items"]), 4, **i32), n_items=tail["n_items"],
                                 cpi=tail.get("cpi", 1),
                                 workspace=self.workspace)

    def set_processor_inputs(self, eos_min=None, prompt_tail=None, prompt_ids=None):
        """Per-call data of the processor stage into the runner's static buffers (a captured step points at them)."""
        if self.eos_min is not None:
            self.eos_min.copy_(eos_min)
        if self.prompt_tail is not None:
            self.prompt_tail.copy_(prompt_tail)
        if self.prompt_ids is not None:
            self.prompt_ids.fill_(-1)                                   # LEFT-padded with -1, like an HF batch
            self.prompt_cols = int(prompt_ids.shape[1])                 # the longest prompt of THIS call (0: inputs_embeds prompts)
            if prompt_ids.shape[1]:
                self.prompt_ids[:, -prompt_ids.shape[1]:].copy_(prompt_ids)

    def sample_tail(self, v, c, d, out_scores, return_scores, n_top=0, status_out=None, step=0, step_ptr=None):
        """The reference's per-step tail (vcd_sample.py:185-207) with its `logits_processor` stage: EOS floor and stop-word forcing
        inside the fused kernel; a repetition penalty or Python processors through the contrast-only / plain split (the scores
        row is materialised between the two launches, as in the generic loop)."""
        t, pr = self.tail, self.proc
        kw = {}
        if self.eos_min is not None:
            kw.update(eos_min_step=self.eos_min, step=step, step_ptr=step_ptr)
        if self.force is not None:
            ops.stop_words_match(pr["stop"], self.prompt_tail, self.gen, step=step, step_ptr=step_ptr, out=self.force)
            kw.update(force_eos=self.force, force_eos_id=pr["stop"].eos_token_id)
        eos_kw = dict(eos_ids=t["eos_t"], pad_id=t["pad"], unfinished=self.unfinished) if t["eos_t"] is not None else {}
        if pr.get("rep") is not None or pr.get("python"):
            if c is not None:
                x = contrast_sample(v, c, d, alpha=t["alpha"], beta=t["beta"], no_sample=True, return_scores=True).scores
            else:
                x = v.clone()
            if pr.get("rep") is not None:
                from .sampling import GPU_SCALAR_SEMANTICS
                ops.repetition_penalty_(x, pr["rep"], self.prompt_ids, self.gen, step=step, step_ptr=step_ptr,
                                        reciprocal=GPU_SCALAR_SEMANTICS)
            if pr.get("python"):                 # eager only (step is a host integer here): HF-style callables see what HF would hand them -
                # exactly max(prompt length) prompt columns (none for inputs_embeds prompts), shorter rows LEFT-padded with the pad id
                pad = t["pad"] if t["pad"] is not None else 0
                L = self.prompt_cols
                pids = self.prompt_ids[:, self.prompt_ids.shape[1] - L:] if L > 0 else self.prompt_ids[:, :0]
                ids = torch.cat([torch.where(pids < 0, pad, pids), self.gen[:, :step]], 1)
                for f in pr["python"]:
                    x = f(ids, x)
            v, c, d = x, None, None
        return contrast_sample(v, c, d, alpha=t["alpha"], beta=t["beta"], warp=t["warp"], out_tokens=self.tok, out_scores=out_scores,
                               return_scores=return_scores, pick_argmax=t["greedy"], seed=0, offset=0, offset_ptr=self.ctr, n_top=n_top,
                               status_out=status_out, **eos_kw, **kw)

    def reset(self, ctr0):
        self.unfinished.fill_(1)
        self.status.zero_(); self.status0.zero_()
        self.step_idx.fill_(0)
        self.ctr.fill_(ctr0)

    def load(self, pos, cpos, slot, rows):
        dev = self.pos.device
        if self.grouping is not None:
            groups, members = group_rows_by_prefix(rows)
            items = ops.prefix_work_items(groups, self.grouping["cpi"])
            assert len(groups) == self.grouping["n_groups"] and len(items) == self.grouping["n_items"]
            p_, c_, s_, r_, g_, i_, m_ = h2d_int32(dev, pos, cpos, slot, rows, groups, items, members)
            self.grouping["groups"].copy_(g_)
            self.grouping["items"].copy_(i_)
            self.grouping["group_rows"][: len(members)].copy_(m_)
        else:
            p_, c_, s_, r_ = h2d_int32(dev, pos, cpos, slot, rows)
        self.pos.copy_(p_); self.cpos.copy_(c_); self.slot.copy_(s_); self.rows.copy_(r_)
        self.gen[:, 0] = self.tok
        self.step_idx.fill_(1)
        self.ctr += 1

    def adopt(self, old: "_DecodeRunner", q_idx: torch.Tensor, n_new: int):
        """Continue `old`'s decoding with the questions q_idx (int64 device tensor, ascending) only: their rows (branch-major in both
        runners) take the slots 0 .. len - 1 of the repacked own pools, in row order.  Device -> device, no sync."""
        Qo, Qn, nb = old.Q, self.Q, self.nb
        rows_idx = torch.cat([q_idx + b * Qo for b in range(nb)])
        self.pos.copy_(old.pos[rows_idx]); self.cpos.copy_(old.cpos[rows_idx])
        self.rows.copy_(old.rows[rows_idx])
        new_slot = torch.arange(nb * Qn, dtype=torch.int32, device=self.pos.device)
        self.slot.copy_(new_slot)
        self.rows[:, 0] = new_slot
        self.tok.copy_(old.tok[q_idx])
        self.unfinished.copy_(old.unfinished[q_idx])
        self.gen[:, :n_new] = old.gen[q_idx, :n_new]
        self.step_idx.copy_(old.step_idx); self.ctr.copy_(old.ctr)
        self.status.copy_(old.status[q_idx]); self.status0.copy_(old.status0[q_idx])
        # the processor stage's per-question state (EOS floor, stop-word prompt tails, repetition-penalty history) moves with its question
        if self.eos_min is not None:
            self.eos_min.copy_(old.eos_min[q_idx])
        if self.prompt_tail is not None:
            self.prompt_tail.copy_(old.prompt_tail[q_idx])
        if self.prompt_ids is not None:
            self.prompt_ids.copy_(old.prompt_ids[q_idx])
            self.prompt_cols = old.prompt_cols
        return old.slot[rows_idx].long()                           # the old slots of the kept rows, in the new slot order

    def admit(self, q_idx: torch.Tensor, tok, unfinished, status0, pos, cpos, rows, pad: int):
        """Admission mode: the question slots q_idx (int64 device tensor) start NEW questions whose first token `tok` was just sampled from
        their prefill; pos / cpos / rows: the decode state of their nb x len(q_idx) rows, branch-major.  Contents of the static buffers
        change, their addresses do not: the captured step keeps replaying."""
        Q, nb, k = self.Q, self.nb, int(q_idx.numel())
        r_idx = torch.cat([q_idx + b * Q for b in range(nb)])
        p_, c_, w_ = h2d_int32(self.pos.device, pos, cpos, rows)
        self.pos.index_copy_(0, r_idx, p_); self.cpos.index_copy_(0, r_idx, c_); self.rows.index_copy_(0, r_idx, w_)
        self.tok.index_copy_(0, q_idx, tok)
        self.unfinished.index_copy_(0, q_idx, unfinished)
        self.status.index_fill_(0, q_idx, 0)
        self.status0.index_copy_(0, q_idx, status0)
        fresh = torch.full((k, self.gen.shape[1]), pad, dtype=torch.long, device=self.gen.device)
        fresh[:, 0] = tok
        self.gen.index_copy_(0, q_idx, fresh)
        self.s0.index_copy_(0, q_idx, (self.step_idx - 1).expand(k))        # the next step writes column 1 of these slots

    def adopt_slots(self, old: "_DecodeRunner", q_idx: torch.Tensor):
        """Admission mode, tail of a list: continue `old`'s decoding with its question slots q_idx only.  Own-KV slots are addressed per row
        (`slot`), so nothing moves in the pools - the smaller step just lists the surviving rows."""
        Qo, nb = old.Q, self.nb
        rows_idx = torch.cat([q_idx + b * Qo for b in range(nb)])
        self.pos.copy_(old.pos[rows_idx]); self.cpos.copy_(old.cpos[rows_idx]); self.slot.copy_(old.slot[rows_idx]); self.rows.copy_(old.rows[rows_idx])
        self.tok.copy_(old.tok[q_idx]); self.unfinished.copy_(old.unfinished[q_idx]); self.gen.copy_(old.gen[q_idx]); self.s0.copy_(old.s0[q_idx])
        self.step_idx.copy_(old.step_idx); self.ctr.copy_(old.ctr)
        self.status.copy_(old.status[q_idx]); self.status0.copy_(old.status0[q_idx])

    def body(self, kv):
        t, Q, nb = self.tail, self.Q, self.nb
        self.tokens_rows.view(nb, Q).copy_(self.tok[None].expand(nb, Q))        # same new token for every branch of a question
        logits = self.eng.lm.decode_step(self.tokens_rows, self.pos, self.cpos, self.slot, self.rows, kv, self.grouping,
                                         workspace=self.workspace)
        v, c, d = logits[:Q], None, None
        if t["contrast"]:
            if t["is_vcd"]:
                c = v                                                           # quirk #1: the cd branch runs on the main cache -> c == v
                d = logits[Q:2 * Q] if t["both"] else None
            else:
                c = logits[Q:2 * Q]
                d = logits[2 * Q:3 * Q] if nb == 3 else None
        if self.proc.get("python"):
            self.sample_tail(v, c, d, self.scores_buf, False, status_out=self._st, step=int(self.step_idx.item()))
        else:
            self.sample_tail(v, c, d, self.scores_buf, False, status_out=self._st, step_ptr=self.step_idx)
        self.status |= self._st
        if self.s0 is None:
            self.gen.index_copy_(1, self.step_idx, self.tok[:, None])
            self.pos += 1
            self.cpos += 1
            self.rows[:, 1] += 1
            if t["eos_t"] is not None and self.grouping is None and self.scores_buf is None:
                # a question that has emitted EOS only produces pad tokens from here on (vcd_sample.py:260) and nobody reads its scores: its
                # rows stop attending their context - ONE key - until the batch ends or they are retired (K / V bytes of a step follow the
                # LIVE rows; the grouped pass keeps its group tables, and per-step score rows stay the reference's)
                dead = (self.unfinished == 0).repeat(nb)
                self.rows[:, 1].masked_fill_(dead, 1)
                self.rows[:, 3].masked_fill_(dead, 0)
        else:
            # per-slot column; a slot whose answer reached max_new tokens is finished like one that emitted EOS; finished slots (pad tokens
            # from the kernel) write the slack column and do not move: they rewrite ONE KV position and attend a fixed context until a
            # waiting question takes the slot
            col = (self.step_idx - self.s0).clamp_(max=self.max_new)
            self.gen.scatter_(1, col[:, None], self.tok[:, None])
            self.unfinished.mul_((col + 1 < self.max_new).long())
            alive = self.unfinished.to(torch.int32).repeat(nb)
            self.pos += alive
            self.cpos += alive
            self.rows[:, 1] += alive
            # ... and stop attending their context: a finished row reads ONE key until its slot is taken (a third of the K / V bytes of a
            # step were those of finished rows near the end of a list)
            dead = alive == 0
            self.rows[:, 1].masked_fill_(dead, 1)
            self.rows[:, 3].masked_fill_(dead, 0)
        self.step_idx += 1
        self.ctr += 1

    def _state(self):
        return [self.tokens_rows, self.pos, self.cpos, self.slot, self.rows, self.tok, self.unfinished, self.gen, self.step_idx, self.ctr,
                self.status, self._st]

    def step(self, kv):
        if self.graph is None and self.eng.use_graph and not self._tried and not self.proc.get("python"):
            self._tried = True
            saved = [x.clone() for x in self._state()]
            side = torch.cuda.Stream(device=self.eng.device)
            side.wait_stream(torch.cuda.current_stream(self.eng.device))
            with torch.cuda.stream(side):                       # warm-up: library handles, allocator pools, attention workspace
                for _ in range(2):
                    self.body(kv)
                    for x, sv in zip(self._state(), saved):     # the warm-ups only scribble the KV row of the CURRENT position,
                        x.copy_(sv)                             # which the real step rewrites before reading it
            torch.cuda.current_stream(self.eng.device).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            # a cyclic-GC run inside the capture may destroy an older engine's graph (hipGraphDestroy / pool release are
            # illegal while a stream is capturing -> abort): collect before, keep the collector off during the capture
            gc.collect()
            gc_was_on = gc.isenabled()
            gc.disable()
            try:
                with torch.cuda.graph(g, stream=side):              # the warm-up stream: its GEMM workspace exists (ops._gemm_workspace)
                    self.body(kv)
            finally:
                if gc_was_on:
                    gc.enable()
            self.graph = g
        if self.graph is not None:
            self.graph.replay()
        else:
            self.body(kv)


class VddLlavaEngine:
    """model.generate()-compatible surface (llava_calibrate.py:161-177) over the native kernels."""

    # generate() kwargs of the reference's drivers that have no effect on this path: KV caching is always on, attention maps /
    # hidden states are never materialised, masks are implied by the ragged prompts, length_penalty only acts on beam search
    LAUNCH_WINDOW = 16        # decode steps between host waits (see the decode loop): bounds the kernel packets queued at any time

    IGNORED_GENERATE_KWARGS = frozenset({"use_cache", "output_hidden_states", "attention_mask", "length_penalty", "synced_gpus", "use_image"})

    def __init__(self, cfg: LlavaConfig | str = "llava-1.5-7b", weights: Optional[LlavaWeights] = None, device="cuda:0",
                 seed: int = 0, max_questions: int = 64, t_max: int = 0, use_graph: bool = True, lm_head_gain: float = 1.0,
                 dtype: Optional[torch.dtype] = None):
        """dtype: torch.bfloat16 (default; BASELINE config #2) or torch.float16 - the dtype every released driver of the reference loads
        its checkpoint in (builder.py:40, llava_calibrate.py:163).  With `weights` given, their dtype is the engine's."""
        self.cfg = preset(cfg) if isinstance(cfg, str) else cfg
        self.device = torch.device(device)
        if weights is not None and dtype is not None and weights.dtype != dtype:
            raise ValueError(f"weights are {weights.dtype}, engine asked for {dtype}")
        self.dtype = weights.dtype if weights is not None else (dtype if dtype is not None else torch.bfloat16)
        if self.dtype not in (torch.bfloat16, torch.float16):
            raise ValueError(f"model dtype must be torch.bfloat16 or torch.float16 (got {self.dtype}): the kernels are built for those two")
        self.w = weights if weights is not None else LlavaWeights.random(self.cfg, self.device, seed, lm_head_gain=lm_head_gain, dtype=self.dtype)
        self.vit = VisionTower(self.w)
        self.vit.use_graph = use_graph
        self.lm = LanguageModel(self.w)
        self.max_q, self.t_max, self.use_graph = max_questions, t_max, use_graph
        # Open-ended answers (LLaVA-Bench: 20 - 1,000 tokens, llava_sampling.py:100-116): rows that emitted EOS leave the batch at the
        # next host check once half of it is done (a re-captured step costs ~0.2 s: tools, bench `llava_bench_eos`), and the own KV pools start at `kv_chunk` new tokens per row and grow by half
        # when the longest live row gets near the end - instead of n_rows x (suffix + max_new_tokens) up front and a batch that decodes
        # at full width until its slowest member stops.  (retire=False: the static form.)
        self.retire, self.retire_fraction, self.kv_chunk = True, 0.5, 128
        self.retire_min_rows = 0          # > 0: no retirement below this many live rows (the pools still grow)
        # decode attention reads each shared prompt prefix once per GROUP of rows (K/V tiles staged in LDS)
        self.group_attention = True
        self._kv: Optional[KVCache] = None          # the pools of the most recent call
        self._kvs: Dict[bool, KVCache] = {}
        self._feat_cache: Dict[int, torch.Tensor] = {}
        self._prefill_cache: Optional[dict] = None           # generate(reuse_prefill=True): plan, pools and step-0 logits of the last such call
        self._graphs: dict = {}
        # a list: every generate() appends its `stats`, with three HIP events (call start, first token sampled, last step issued) under
        # "events" - how bench.py splits a driver's wall time into prefill and decode steps without adding a sync (`call_timing`)
        self.call_log: Optional[list] = None

    # -- plumbing ---------------------------------------------------------------------------
    def kv(self, n_pre, t_pre, n_own, t_own, frag_only=False):
        """The KV pools for a call, one cache per prefix-pool form (KVCache.frag_only): a driver alternates calls that group their
        decode rows (the image questions) with calls that do not (a handful of rows), and each keeps its pools - and the decode
        graphs captured over them - across calls; a cache only grows."""
        # prefix slots are whole 64-key chunks (the fragment image); own slots only need whole 16-key groups: 92 own tokens take
        # 96 rows, not 128 (a quarter of the own pool, 26 GB at 768 questions)
        t_pre, t_own = (max(t_pre, 64) + 63) // 64 * 64, (max(t_own, 64) + 15) // 16 * 16
        cur = self._kvs.get(frag_only)
        if cur is None or not cur.fits(n_pre, t_pre, n_own, t_own, frag_only):
            self._kvs.pop(frag_only, None)
            for k in [k for k, r in self._graphs.items() if r.kv is cur]:     # captured steps point into the pools being replaced
                self._graphs.pop(k)
            grow = lambda a, b: max(a, b)
            if cur is not None:
                n_pre, t_pre, n_own, t_own = grow(n_pre, cur.n_pre), grow(t_pre, cur.t_pre), grow(n_own, cur.n_own), grow(t_own, cur.t_own)
            if self._kv is cur:
                self._kv = None
            del cur
            need = KVCache.bytes_needed(self.cfg.lm, n_pre, t_pre, n_own, t_own, self.dtype, frag_only)
            other = self._kvs.get(not frag_only)
            if other is not None and self.device.type == "cuda":
                free, _total = torch.cuda.mem_get_info(self.device)
                if need > free + torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device):
                    # both forms do not fit side by side: the other one (and its graphs) goes
                    for k in [k for k, r in self._graphs.items() if r.kv is other]:
                        self._graphs.pop(k)
                    self._kvs.pop(not frag_only)
                    if self._kv is other:
                        self._kv = None
                    del other
                    gc.collect()
                    torch.cuda.empty_cache()
            self._kvs[frag_only] = KVCache(self.cfg.lm, n_pre, t_pre, n_own, t_own, self.device, self.dtype, frag_only=frag_only)
        self._kv = self._kvs[frag_only]
        return self._kv

    # Distinct images per vision-tower forward.  The tower's GEMMs have K = 1,024 / 4,096 and N = 1,024 ... 4,096: at 16 images (9,232
    # rows) they are 148 - 592 tiles of 256 x 256 on 256 CUs and ran at 0.5 PF/s in the bench trace; at 64 images every product is
    # 2 - 9 full rounds: 0.761 -> 0.568 ms per image (0.524 at 128; tools/vit_batch_probe.py, profiles/r05_vit_batch_probe.jsonl).  Chunks
    # of VisionTower.GRAPH_SIZES images replay a captured graph; larger ones are GPU-bound issued from Python (same time either way).
    VIT_CHUNK = 64

    def image_features(self, images: Sequence[torch.Tensor], keys: Optional[Sequence[int]] = None) -> List[torch.Tensor]:
        """ViT + projector per DISTINCT image (POPE: 6 questions share one image)."""
        if "v.patch" not in self.w.t:
            raise ValueError("this engine was built from a language model only (hf_adapter.attach_lm_engine / attach_blip_engine / "
                             "attach_qwen_engine): it takes inputs_embeds or text ids, not images")
        if keys is not None:                     # caller-supplied identities: cache persists across generate() calls
            keys, cache = list(keys), self._feat_cache
        else:                                    # same storage within this call = same image
            keys, cache = [(im.data_ptr(), tuple(im.shape)) for im in images], {}
        todo, seen = [], set()
        for k, im in zip(keys, images):
            if k not in cache and k not in seen:
                todo.append((k, im)); seen.add(k)
        C_ = self.VIT_CHUNK
        for i in range(0, len(todo), C_):
            chunk = todo[i:i + C_]
            ims = [im.reshape(im.shape[-3:]) for _, im in chunk]
            staged = self._stage_host_images(ims, i // C_)
            feats = self.vit(staged if staged is not None else torch.stack(ims))
            if staged is not None:
                self._pin_done[(i // C_) % 2].record(torch.cuda.current_stream(self.device))
            for (k, _), f in zip(chunk, feats):
                cache[k] = f
        return [cache[k] for k in keys]

    def _stage_host_images(self, ims, n_chunk):
        """Host images (the reference's drivers hand over CPU tensors: llava_calibrate.py:146-147) go through two PINNED staging
        buffers: stacked straight into one (a host memcpy), uploaded by ONE asynchronous copy while the host stacks the next
        chunk into the other.  From pageable memory every chunk's upload is synchronous and goes through the runtime's bounce
        buffers: 0.4 s per 128 images, a tenth of the batch time (`pcie_inclusive` on the bench line)."""
        if not ims or ims[0].is_cuda or any(im.is_cuda or im.dtype != ims[0].dtype or im.shape != ims[0].shape for im in ims):
            return None
        key = (ims[0].dtype, tuple(ims[0].shape), self.VIT_CHUNK)
        if getattr(self, "_pin_key", None) != key:
            if getattr(self, "_pin_used", None):                 # uploads still reading the buffers about to be dropped
                for b_, used in enumerate(self._pin_used):
                    if used:
                        self._pin_done[b_].synchronize()
            self._pin = [torch.empty((self.VIT_CHUNK,) + key[1], dtype=key[0], pin_memory=True) for _ in range(2)]
            self._pin_done = [torch.cuda.Event() for _ in range(2)]
            self._pin_key = key
            self._pin_used = [False, False]
        b = n_chunk % 2
        if self._pin_used[b]:
            self._pin_done[b].synchronize()          # the upload that last read this buffer
        self._pin_used[b] = True
        out = self._pin[b][:len(ims)]
        torch.stack(ims, out=out)
        return out

    def clear_image_cache(self):
        self._feat_cache.clear()

    @staticmethod
    def call_timing(stats: dict) -> dict:
        """Prefill (vision tower + prompt passes + first token) and decode-loop time of one logged call, from the events of `call_log`
        (synchronises on the last one).  decode_ms covers steps - 1 decode steps (the first token comes out of the prefill)."""
        e0, e1, e2 = stats["events"]
        e2.synchronize()
        return {"prefill_ms": e0.elapsed_time(e1), "decode_ms": e1.elapsed_time(e2), "decode_steps": max(int(stats.get("steps_run", 1)) - 1, 0)}

    # -- generate -------------------------------------------------------------------------------
    def generate(self, *args, **kwargs) -> "GenerateOutput":
        """See `_generate`.  Runs it with the cyclic garbage collector paused: the prefill issues ~2,000 launches from
        Python while holding a few thousand small planning objects, and a generation-2 collection landing in the middle
        stalls the launch queue for 50-200 ms (rocprofv3 kernel trace of `bench.py`: prefill GPU-busy time 740 ms in
        every call, wall 764-1,061 ms with one or two such holes) - reference counting still frees everything promptly."""
        was_on = gc.isenabled()
        gc.disable()
        try:
            return self._generate(*args, **kwargs)
        finally:
            if was_on:
                gc.enable()

    @torch.no_grad()
    def _generate(self, input_ids: Sequence[torch.Tensor] | torch.Tensor, images=None, images_cd=None, image_keys=None,
                 cd_alpha: Optional[float] = None, cd_beta: Optional[float] = None, use_dd: bool = False,
                 use_dd_unk: bool = False, do_sample: bool = True, temperature: Optional[float] = None,
                 top_p: Optional[float] = None, top_k: Optional[int] = None, max_new_tokens: int = 64,
                 eos_token_id=None, pad_token_id: Optional[int] = None, output_scores: bool = False,
                 return_dict_in_generate: bool = True, cd_greedy: bool = False, n_top: int = 0, seed: Optional[int] = None,
                 share_prefix: bool = True, sync_every: int = 8, inputs_embeds=None, min_new_tokens: Optional[int] = None,
                 min_length: Optional[int] = None, stop_words_ids=None, repetition_penalty: Optional[float] = None,
                 logits_processor=None, max_length: Optional[int] = None, num_beams: Optional[int] = None,
                 num_return_sequences: Optional[int] = None, embeds_prefix=None, streamer=None, output_attentions: bool = False,
                 branch_priors: bool = False, reuse_prefill: bool = False, **other) -> GenerateOutput:
        """Same kwargs as the reference's model.generate(...) call (llava_calibrate.py:161-177); `input_ids` is a
        list of 1-D id tensors (one per question, each with one -200 image slot) or a [Q, L] tensor; `images` one
        image per question (repeat the SAME tensor for questions about the same image to share its features and
        prompt-prefix KV).  use_cache is accepted and ignored.  output_attentions=True with ONE question (the reference's call,
        llava_calibrate.py:161-177) materialises what the driver reads of it - `['attentions'][0][-1]`, step 0 / last layer, [1, H, T, T]
        of the main branch's prompt (:180-182) - from the prefill's rotated queries and the cached keys (ops.attention_probs); with several
        questions it is accepted without effect and `['attentions']` raises.  attention_mask is accepted when it is all ones (the
        reference's B = 1 calls) and refused when it masks anything.

        embeds_prefix (with inputs_embeds): per prompt `(key, n)` - the caller's promise that prompts with the same key start with the
        same n embedding rows (Qwen-VL: '<img>' + the 256 image slots of one image, shared by its questions): they are prefilled once
        into a shared prefix slot, like [system prompt + image] of the LLaVA path.

        Logits processors (what HF's generate() builds into `logits_processor` for vcd_sample.py:197 / :204, in HF's order):
        repetition_penalty (blip2_vicuna_instruct.py:400), min_length (:397; counts the prompt, which is EMPTY for inputs_embeds
        prompts as in HF), min_new_tokens (MME/run_qwen.py:194), stop_words_ids (Qwen, modeling_qwen.py:1061-1075: forces
        eos_token_id[0] once a stop sequence ends the row), then `logits_processor`: a list of HF-style callables
        `f(input_ids, scores)` run in eager mode on left-padded ids.  max_length (LAVIS passes it instead of max_new_tokens) =
        prompt length + new tokens; the prompt length of inputs_embeds prompts is 0.  Beam search / several return sequences are
        not part of the patched sample() path: refused.  Any other keyword raises TypeError instead of being dropped.

        streamer (HF BaseStreamer: llava/serve/cli.py [ext] passes a TextStreamer): `put(prompt ids)` as HF's generate() does before
        sampling, `put(next token)` after every step (vcd_sample.py:264-265: one device -> host copy per step) and `end()` at the
        end (:299-300).  One question per call, as HF's own streamers require."""
        dev, lm = self.device, self.cfg.lm
        events = None
        if self.call_log is not None and dev.type == "cuda":
            events = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            events[0].record(torch.cuda.current_stream(dev))
        am = other.get("attention_mask")
        if am is not None and not bool(torch.as_tensor(am).ne(0).all()):
            raise ValueError("attention_mask with zeros: a left-padded [Q, L] id tensor would be decoded with its pad tokens as prompt - "
                             "pass a list of per-question id tensors (ragged prompts are batched natively)")
        unknown = sorted(k for k in other if k not in self.IGNORED_GENERATE_KWARGS)
        if unknown:
            raise TypeError(f"generate() got unexpected keyword argument(s) {unknown}: not implemented by VddLlavaEngine "
                            f"(accepted without effect: {sorted(self.IGNORED_GENERATE_KWARGS)})")
        if streamer is not None and (len(input_ids) if inputs_embeds is None else len(inputs_embeds)) != 1:
            raise ValueError("streamer: one question per generate() call (HF's TextStreamer only supports batch size 1)")
        if num_beams not in (None, 1) or num_return_sequences not in (None, 1):
            raise ValueError("num_beams / num_return_sequences > 1: the reference patches sample() only (vcd_sample.py:325-326); "
                             "beam search never reaches contrastive decoding")
        if inputs_embeds is not None:
            # LAVIS / InstructBLIP call shape (blip2_vicuna_instruct.py:380-410): the prompt arrives as embeddings
            # [T, d] per question (Q-Former output ++ text embeddings) and `images_cd` holds the noisy-image EMBEDDINGS,
            # which modeling_llama.py:778-782 feeds as inputs_embeds of the cd branch at step 0.
            # use_dd / use_dd_unk with a prompt that carries NO -200 placeholder is the reference's Qwen-VL case (SURVEY A.3 #4:
            # its ids never contain IMAGE_TOKEN_INDEX, vcd_sample.py:154-160 change nothing): the image-free branches re-run the
            # SAME inputs on their own KV caches, c ~ v, and the contrast reduces to the beta-mask.  Reproduced as such: the extra
            # branches get the main branch's embeddings (the dual / triple forward pass is really executed).
            emb_main = [e.reshape(-1, lm.d) for e in (inputs_embeds if not torch.is_tensor(inputs_embeds) else list(inputs_embeds))]
            emb_cd = None
            if images_cd is not None:
                emb_cd = [e.reshape(-1, lm.d) for e in (images_cd if not torch.is_tensor(images_cd) else list(images_cd))]
            input_ids = [torch.zeros(0, dtype=torch.long) for _ in emb_main]
            images = None
        ids_list = [r for r in input_ids] if torch.is_tensor(input_ids) else list(input_ids)
        ids_list = [r.reshape(-1).tolist() for r in ids_list]
        Q = len(ids_list)
        prompt_lens = [0] * Q if inputs_embeds is not None else [len(r) for r in ids_list]     # HF's input_ids length per row
        if max_length is not None:
            if len(set(prompt_lens)) != 1:
                raise ValueError("max_length with prompts of different lengths: pass max_new_tokens")
            max_new_tokens = int(max_length) - prompt_lens[0]
            if max_new_tokens < 1:
                raise ValueError(f"max_length {max_length} leaves no room behind a prompt of {prompt_lens[0]} tokens")
        # the vision tower needs nothing of the planning below: its launches go out first, so the host-side validation / planning
        # of ~800 prompts (20-30 ms of Python) runs under its GPU time instead of in front of it
        # reuse_prefill: a sweep over sampling settings (the reference's scripts run 51 of them over the same questions, MME/run_llava.py:281-318)
        # asks for the SAME prompts again - only the warpers / seed / alpha / beta differ, and none of them enters the prefill.  The call keeps
        # its plan, pools and step-0 logits; the next reuse_prefill call with the same prompt and image tensors (the caller's promise: unchanged),
        # the same branches and no more new tokens skips the vision tower and the prefill and decodes again from that state (a decode only writes
        # own-KV positions behind the prompt).  Not with the VCD branch (fresh noise per call); such calls never retire rows.
        hit, rkey = None, None
        if reuse_prefill and not output_attentions:
            eff = (False, False) if not do_sample else (bool(use_dd), bool(use_dd_unk))
            if images_cd is None or not do_sample:
                ims = None if images is None else tuple((images[i].data_ptr(), tuple(images[i].shape)) for i in range(Q))
                embs = None if inputs_embeds is None else tuple((e.data_ptr(), tuple(e.shape)) for e in emb_main)
                rkey = (tuple(map(tuple, ids_list)), ims, embs, tuple(embeds_prefix) if embeds_prefix is not None else None, eff, share_prefix,
                        self.two_level_prefix, self.share_repeated_rows, self.group_attention, ops.GEMM_BATCH_INVARIANT, str(self.dtype))
                c_ = self._prefill_cache
                if c_ is not None and c_["key"] == rkey and self._kvs.get(c_["frag"]) is c_["kv"] and max_new_tokens <= c_["max_new"]:
                    hit = c_
        feats = None
        if images is not None and hit is None:
            imgs = [images[i] for i in range(Q)] if torch.is_tensor(images) else list(images)
            feats = self.image_features(imgs, image_keys)
        for q_, r in enumerate(ids_list):                     # ids index the embedding table on the device: validate them here
            n_slot = sum(1 for t_ in r if t_ == IMAGE_TOKEN_INDEX)
            if n_slot > 1:
                raise ValueError(f"prompt {q_}: {n_slot} image placeholders (-200); the LLaVA path splices exactly one image per prompt")
            bad = [t_ for t_ in r if t_ != IMAGE_TOKEN_INDEX and not (0 <= t_ < lm.vocab)]
            if bad:
                raise ValueError(f"prompt {q_}: token id {bad[0]} outside [0, {lm.vocab}) (only -200 marks the image slot)")
        if inputs_embeds is None and images is None and any(IMAGE_TOKEN_INDEX in r for r in ids_list):
            raise ValueError("input_ids contain the image placeholder (-200) but no `images` were given")
        alpha = cd_alpha if cd_alpha is not None else 0.5                                     # vcd_sample.py:188
        beta = cd_beta if cd_beta is not None else 0.1                                        # :189
        use_cd = images_cd is not None
        if not do_sample:                  # reference: greedy_search is not patched -> no contrast (A.3 #5)
            use_cd = use_dd = use_dd_unk = False
            cd_greedy, temperature, top_p, top_k = True, None, None, None
        contrast = use_cd or use_dd or use_dd_unk
        if isinstance(eos_token_id, int):
            eos_token_id = [eos_token_id]
        if eos_token_id is not None and pad_token_id is None:
            raise ValueError("If `eos_token_id` is defined, make sure that `pad_token_id` is defined.")   # :258-259
        warp = WarpSpec(temperature=temperature, top_k=top_k, top_p=top_p)
        proc, proc_key = self._processor_config(prompt_lens, ids_list, eos_token_id, min_new_tokens, min_length, stop_words_ids,
                                                repetition_penalty, logits_processor, max_new_tokens)

        # ---- branches: (name, per-question token lists with image slot handling) ------------
        feats_cd = None
        if use_cd and inputs_embeds is None:
            imgs_cd = [images_cd[i] for i in range(Q)] if torch.is_tensor(images_cd) else list(images_cd)
            # the noised copies differ per question (fresh noise, llava_calibrate.py:152-155): no feature cache, but full tower chunks
            # instead of one launch chain per question
            feats_cd = []
            for i0 in range(0, Q, self.VIT_CHUNK):
                chunk = [im.reshape(im.shape[-3:]).to(dev) for im in imgs_cd[i0:i0 + self.VIT_CHUNK]]
                feats_cd += list(self.vit(torch.stack(chunk)))
        if hit is not None:
            branches = hit["branches"]
        elif inputs_embeds is not None:
            main_dev = [e.to(dev, self.dtype) for e in emb_main]
            branches = [("main", ids_list, main_dev)]
            if use_cd:
                branches.append(("cd", ids_list, [e.to(dev, self.dtype) for e in emb_cd]))
            elif use_dd_unk:
                branches.append(("unk", ids_list, main_dev))
            elif use_dd:
                branches.append(("none", ids_list, main_dev))
            if use_dd and use_dd_unk:
                branches.append(("none", ids_list, main_dev))
        else:
            branches = [("main", ids_list, feats)]
        if hit is not None:
            pass
        elif use_cd and inputs_embeds is None:
            branches.append(("cd", ids_list, feats_cd))                                       # :148-150, takes precedence
        elif inputs_embeds is not None:
            pass
        elif use_dd_unk:
            branches.append(("unk", [[0 if t == IMAGE_TOKEN_INDEX else t for t in r] for r in ids_list], None))   # :154-155
        elif use_dd:
            branches.append(("none", [[t for t in r if t != IMAGE_TOKEN_INDEX] for r in ids_list], None))         # :157-160
        if use_dd and use_dd_unk and inputs_embeds is None and hit is None:
            branches.append(("none", [[t for t in r if t != IMAGE_TOKEN_INDEX] for r in ids_list], None))         # :171-177
        nb = len(branches)

        # ---- plan prefill: split every (branch, question) sequence into shared prefix + own suffix -----
        n_img_tok = self.cfg.vision.n_patches
        if embeds_prefix is not None and (inputs_embeds is None or len(embeds_prefix) != Q):
            raise ValueError("embeds_prefix goes with inputs_embeds: one (key, n_rows) per prompt")
        if hit is not None:
            plan = hit["plan"]
        else:
            plan = self._plan(branches, n_img_tok, share_prefix, embeds_only=inputs_embeds is not None, embeds_prefix=embeds_prefix)
            if self.two_level_prefix and inputs_embeds is None and not (output_attentions and Q == 1):
                self._split_system_prompt(plan)
        # which rows decode, and whether their shared prefixes are attended through the grouped MFMA pass - decided BEFORE the prefill: it
        # settles the form the prefix K/V are kept in (KVCache: fragment image only, or row-major only)
        seg = plan["suffix"]
        keep = [i for i, (name, _, _) in enumerate(branches) if not (use_cd and name == "cd")]
        sel = [b_ * Q + q for b_ in keep for q in range(Q)]
        dec_rows = [[seg[i]["slot"], seg[i]["pos0"] + seg[i]["T"] + 1, seg[i]["pslot"], seg[i]["plen"]] for i in sel]
        grp, _members = group_rows_by_prefix(dec_rows) if (self.group_attention and not ops.GEMM_BATCH_INVARIANT) else ([], [])
        if grp and (not grouping_pays(grp, dec_rows) or (len(dec_rows) <= ops.FUSED_ATTN_MAX_M and lm.head_dim == 128)):
            grp = []      # (up to 16 rows the one-launch RoPE + KV write + attention kernel beats the three launches of the grouped
                          #  path although it reads a shared prefix once per row: tools/small_batch_attn_probe.py, +4 ... 9 %)
        # retirement / growth of the own pools needs an EOS to retire on, per-question state that lives only in the runner's row order
        # (no per-step scores rows, no streamer) and the ungrouped attention (slots are renumbered)
        suffix_max = max(s_["T"] for s_ in plan["suffix"])
        # ... and, for deterministic decodes (cd_greedy / top_k = 1), only in batch-invariant mode: with the tuned forms a survivor's low-order bits
        # change as the batch shrinks through the kernel regimes, where the reference keeps the full batch until every row has emitted EOS
        deterministic = bool(cd_greedy or top_k == 1)
        # (round 6: the in-kernel processors - EOS floor of min_new_tokens / min_length, stop words, repetition penalty - are per-question state
        #  that `adopt` carries along; only HF-style Python callables, which see the whole left-padded batch, still pin the batch)
        retire = bool(self.retire and eos_token_id is not None and not output_scores and streamer is None and not (proc and proc.get("python"))
                      and not grp and max_new_tokens > self.kv_chunk and (ops.GEMM_BATCH_INVARIANT or not deterministic) and not reuse_prefill)
        own_cap = min(max_new_tokens, self.kv_chunk) if retire else max_new_tokens
        kv = self.kv(len(plan["prefix"]), max([s_.get("full_T", s_["T"]) for s_ in plan["prefix"]] + [0]), len(plan["suffix"]),
                     suffix_max + own_cap, frag_only=bool(grp))
        if hit is not None and (kv is not hit["kv"] or bool(grp) != hit["frag"]):
            raise RuntimeError("reuse_prefill: the pools of the kept prefill were replaced")       # (cannot happen: same plan, no more new tokens)
        if hit is None and self._prefill_cache is not None and self._prefill_cache["kv"] is kv:
            self._prefill_cache = None                         # this call's prefill overwrites the pools the kept state lives in
        if plan["max_len"] + max_new_tokens > self.cfg.lm.max_pos:
            raise ValueError(f"prompt ({plan['max_len']} positions) + max_new_tokens ({max_new_tokens}) exceed the rotary table "
                             f"(max_pos = {self.cfg.lm.max_pos}): lower max_new_tokens or build the engine with a larger LMConfig.max_pos")
        stats = {"n_rows": nb * Q, "prefill_tokens": 0 if hit is not None else plan["prefill_tokens"], "unshared_prefill_tokens": plan["unshared_tokens"],
                 "prefill_reused": hit is not None}
        # the K / V a decode step reads at step 0 (+ one token per row and step after it): per row (SURVEY 8d's sum over rows) and with every
        # shared prefix counted once (the bytes that have to cross the chip at least once per step)
        stats.update(decode_rows=len(dec_rows), ctx_tokens_rows=sum(r[1] for r in dec_rows),
                     ctx_tokens_distinct=sum(r[1] - r[3] for r in dec_rows) + sum({r[2]: r[3] for r in dec_rows if r[3] > 0}.values()))

        want_maps = bool(output_attentions) and Q == 1 and lm.head_dim == 128 and lm.n_layers > 0
        passes, frag_plen = [], None
        if hit is not None:
            logits0 = hit["logits0"]
        else:
            if plan["prefix"]:
                # level 0: the prefixes that continue nothing; level 1: image prefixes behind a shared system prompt (their pass runs second in
                # every layer: it attends the parent's K / V of that layer, and copies them in front of its own rows)
                for level in (0, 1):
                    segs = [s_ for s_ in plan["prefix"] if (s_.get("cpos0", 0) > 0) == bool(level)]
                    if not segs:
                        continue
                    x, pos, cpos, slot, seqs, max_tq = self._pack(segs)
                    pd = dict(x=x, pos=pos, cpos=cpos, slot=slot, seqs=seqs, n_seq=len(segs), max_tq=max_tq, to_prefix_pool=True, keep_q=want_maps)   # K/V only
                    if level:
                        n_sys = segs[0]["cpos0"]
                        pd.update(own_row_offset=n_sys, parent_copy=(h2d_long(dev, [s_["slot"] for s_ in segs]),
                                                                       segs[0]["pslot"], n_sys))
                    passes.append(pd)
                if kv.frag_only:
                    (frag_plen,) = h2d_int32(dev, [s_.get("full_T", s_["T"]) for s_ in plan["prefix"]])
            segs = plan["suffix"]
            x, pos, cpos, slot, seqs, max_tq = self._pack(segs)
            # short suffixes behind shared prefixes: four sequences of one prefix per attention workgroup (ops.flash_packs)
            packs_h = ops.flash_packs([[0, 0, 0, 0, s["pslot"], s["plen"]] for s in segs]) if (max_tq <= 32 and lm.head_dim == 128 and (not ops.GEMM_BATCH_INVARIANT or ops.FLASH_PACKS_IN_INVARIANT_MODE)) else None
            last, last_seqs, packs = h2d_int32(dev, [s["q_row0"] + s["T"] - 1 for s in segs],
                                               [[i, 1, s["pos0"] + s["T"] - 1, s["slot"], s["pslot"], s["plen"]] for i, s in enumerate(segs)],
                                               packs_h if packs_h is not None else [[0, -1, -1, -1]])
            passes.append(dict(x=x, pos=pos, cpos=cpos, slot=slot, seqs=seqs, n_seq=len(segs), max_tq=max_tq, to_prefix_pool=False,
                               last_rows=last.long(), last_seqs=last_seqs, packs=packs if packs_h is not None else None, keep_q=want_maps))
            resid, delta = self.lm.prefill(passes, kv, frag_plen=frag_plen)[-1]
            logits0 = self.lm.logits(resid, delta)                                        # [nb*Q, V], rows ordered branch-major
            if rkey is not None:
                self._prefill_cache = dict(key=rkey, kv=kv, frag=bool(grp), plan=plan, branches=branches, logits0=logits0, max_new=max_new_tokens)
        self.debug_logits0 = logits0
        attn_maps = None
        if want_maps:
            # question 0, main branch = suffix sequence 0 (+ its prefix slot): [prefix rows | suffix rows] of the last layer's rotated queries
            s0 = plan["suffix"][0]
            q_rows = [passes[-1]["q_last"][s0["q_row0"]: s0["q_row0"] + s0["T"]]]
            if s0["plen"] > 0:
                p0 = plan["prefix"][s0["pslot"]]
                q_rows.insert(0, passes[0]["q_last"][p0["q_row0"]: p0["q_row0"] + p0["T"]])
            q_full = torch.cat(q_rows, 0) if len(q_rows) > 1 else q_rows[0].contiguous()
            li = lm.n_layers - 1
            m = ops.attention_probs(q_full, kv.ko[li], (0, q_full.shape[0], 0, s0["slot"], s0["pslot"], s0["plen"]), lm.n_heads, lm.n_kv_heads,
                                    lm.head_dim, k_prefix=kv.kp[li])
            attn_maps = m[None]                                                      # [1, H, T, T] like HF's attention weights
            for p_ in passes:
                p_.pop("q_last", None)
        V = lm.vocab

        # ---- step 0: sample from the prefill logits (eager; also yields the top-n for calibration) -----------
        # device-side Philox counter = (stream, step): graphs are seed-agnostic.  seed=None: the stream id is drawn from torch's
        # default generator, so every generate() call samples fresh numbers (the reference's torch.multinomial advances the
        # generator too) and torch.manual_seed() reproduces a run's random numbers; an explicit seed is a pure function of the seed.
        # (The bf16 logits they are applied to depend, in their last bits, on the GEMM schedule the tuner picked: ops.gemm_choices_export.)
        from .sampling import fresh_offset
        sd = (fresh_offset() if seed is None else int(seed)) & 0x3FFFFFFFFF
        ctr0 = sd << 24
        eos_t = h2d_long(dev, eos_token_id) if eos_token_id is not None else None
        cfgkey = (Q, nb, max_new_tokens, alpha, beta, warp.t, warp.k, warp.p, use_cd, use_dd, use_dd_unk, cd_greedy,
                  tuple(eos_token_id) if eos_token_id is not None else None, pad_token_id, output_scores, ops.GEMM_BATCH_INVARIANT)
        cpi = ops.prefix_chunks_per_item(grp, lm.n_heads)
        n_groups, n_items = len(grp), len(ops.prefix_work_items(grp, cpi))
        cfgkey = cfgkey + (n_groups, n_items, cpi, proc_key)
        run = self._runner(cfgkey, Q, len(keep), max_new_tokens, dict(alpha=alpha, beta=beta, warp=warp, contrast=contrast,
                           is_vcd=use_cd, both=(use_dd and use_dd_unk), greedy=cd_greedy, eos_t=eos_t, pad=pad_token_id,
                           output_scores=output_scores, n_groups=n_groups, n_items=n_items, cpi=cpi, proc=proc), kv)
        run.reset(ctr0)
        if proc:
            run.set_processor_inputs(**self._processor_inputs(proc, prompt_lens, ids_list, min_new_tokens, min_length))
        scores = [] if output_scores else None
        v0 = logits0[:Q]
        c0 = logits0[Q:2 * Q] if contrast else None
        d0 = logits0[2 * Q:3 * Q] if (contrast and nb == 3) else None
        r0 = run.sample_tail(v0, c0, d0, None, output_scores, n_top=n_top, status_out=run.status0, step=0)
        if output_scores:
            scores.append(r0.scores)
        top_prob, top_tok = r0.top_prob, r0.top_tok
        # branch_priors: the step-0 top-n of every image-free branch's OWN distribution under the call's warpers, no contrast - what a separate
        # plain generate() over that branch's ids reports (the POPE driver's `unk` prior pass feeds exactly the ids of the `unk` branch,
        # llava_calibrate.py:59-60 against vcd_sample.py:154-155: the same forward twice in the reference, once here)
        branch_top = None
        if branch_priors and n_top and inputs_embeds is None:
            branch_top = {}
            for bi, (bname, _, _) in enumerate(branches):
                if bi and bname in ("unk", "none") and bname not in branch_top:
                    rb = contrast_sample(logits0[bi * Q:(bi + 1) * Q], None, None, warp=warp, pick_argmax=True, seed=0, offset=0, n_top=n_top)
                    branch_top[bname] = (rb.top_tok, rb.top_prob)
        # the VCD branch has no state of its own after step 0 (quirk #1): only the other branches keep decoding
        run.load(pos=[seg[i]["pos0"] + seg[i]["T"] for i in sel], cpos=[seg[i]["T"] for i in sel], slot=[seg[i]["slot"] for i in sel],
                 rows=dec_rows)
        n_new = 1
        if events is not None:
            events[1].record(torch.cuda.current_stream(dev))
        # retirement state: `alive` = the original question index of every question still in the runner; `master` collects the tokens
        alive = list(range(Q))
        master = None
        if retire:
            master = torch.full((Q, max_new_tokens), pad_token_id if pad_token_id is not None else 0, dtype=torch.long, device=dev)
            stats.update(retire_events=0, rows_at_end=nb * Q, own_capacity=kv.t_own - suffix_max)
        if streamer is not None:
            if inputs_embeds is None:
                streamer.put(torch.tensor([ids_list[0]], dtype=torch.long))
            streamer.put(run.gen[:, 0].cpu())
        # The host runs ahead of the device (a captured step is one graph launch): without an EOS check nothing ever waits, and a 256-token
        # generation used to leave > 100,000 kernel packets queued at once.  The HIP runtime copes (it blocks on a full queue), a tool
        # intercepting the queue does not: rocprofv3 aborted the process / segfaulted at a deterministic dispatch index once ~30 - 60
        # thousand packets were outstanding (tools/rocprof_abort_bisect.py: 64 new tokens fine, 128 x 2 and 256 not, eager fine).
        # So the loop waits for the step launched LAUNCH_WINDOW steps ago before going on: at most 2 x 16 steps in flight, no measurable cost.
        marks = []
        while n_new < max_new_tokens:
            run.step(kv)
            if output_scores:
                scores.append(run.scores_buf.clone())
            n_new += 1
            if n_new % self.LAUNCH_WINDOW == 0:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                marks.append(ev)
                if len(marks) > 1:
                    marks.pop(0).synchronize()
            if streamer is not None:
                streamer.put(run.gen[:, n_new - 1].cpu())
                if eos_t is not None and int(run.unfinished.max().item()) == 0:
                    break
            # the "everybody finished" test costs a host sync: every sync_every steps, and at steps 2 and 4 on the way there (POPE answers
            # are 1-2 tokens: waiting for step 8 would run six decode steps for nobody)
            if eos_t is not None and (n_new % sync_every == 0 or n_new == max_new_tokens or (n_new in (2, 4) and n_new < sync_every)):
                if retire:
                    state = torch.cat([run.unfinished, (run.status | run.status0).ne(0).any().long()[None]]).tolist()    # ONE sync
                    unf, bad = state[:-1], state[-1]
                    if bad or not any(unf):
                        break
                    live_q = [j for j, u in enumerate(unf) if u]
                    own_len = suffix_max + n_new                                         # own rows of the longest live slot (an upper bound)
                    grow = own_len + sync_every + 1 > kv.t_own and n_new < max_new_tokens and kv.t_own < suffix_max + max_new_tokens
                    shrink = len(live_q) <= self.retire_fraction * len(unf) and run.nb * len(live_q) >= self.retire_min_rows
                    if not shrink:
                        live_q = list(range(len(unf)))                                  # growth only: everybody stays
                    if n_new < max_new_tokens and (grow or shrink):
                        # finished questions leave: their tokens go to `master`, the live rows move to the front of new own pools
                        idx_all = torch.tensor(alive, dtype=torch.long, device=dev)
                        master[idx_all, :n_new] = run.gen[:, :n_new]
                        q_idx = torch.tensor(live_q, dtype=torch.long, device=dev)
                        cap = kv.t_own - suffix_max
                        if grow:
                            cap = min(max_new_tokens, max(cap + cap // 2, n_new + 2 * sync_every + 2))
                        Qn = len(live_q)
                        self._graphs = {k_: r_ for k_, r_ in self._graphs.items() if r_.kv is not kv}    # captured steps point into the old pools
                        new_run = _DecodeRunner(self, Qn, run.nb, max_new_tokens, dict(run.tail), kv)
                        old_slots = new_run.adopt(run, q_idx, n_new)
                        kv.repack_own(old_slots, own_len, suffix_max + cap)
                        new_run.workspace = ops.attention_workspace(run.nb * Qn, lm.n_heads, lm.head_dim, kv.t_pre + (kv.t_own + 63) // 64 * 64, dev)
                        for k_ in [k_ for k_ in self._graphs if len(k_) > len(cfgkey) and k_[len(cfgkey)] == "retired"]:
                            self._graphs.pop(k_)                          # the runner of the previous retirement event: nothing replays it again
                        self._graphs[cfgkey + ("retired", Qn, kv.t_own, stats["retire_events"])] = new_run
                        alive = [alive[j] for j in live_q]
                        run = new_run
                        stats["retire_events"] += 1
                        stats.setdefault("retire_log", []).append((n_new, run.nb * Qn, cap))
                        stats["rows_at_end"], stats["own_capacity"] = run.nb * Qn, cap
                    continue
                done_bad = torch.stack([run.unfinished.max() == 0, (run.status | run.status0).ne(0).any()]).tolist()   # ONE sync
                if done_bad[1]:                                 # a row lost every finite score: stop decoding from token -1
                    break
                if done_bad[0]:                                                               # :291, amortised over sync_every steps
                    break
        if events is not None:
            events[2].record(torch.cuda.current_stream(dev))
            stats["events"], stats["steps_run"] = events, n_new
            self.call_log.append(stats)
        if bool((run.status | run.status0).ne(0).any().item()):
            raise RuntimeError("probability tensor contains either `inf`, `nan` or element < 0")   # torch.multinomial, :202
        if streamer is not None:
            streamer.end()
        if retire:
            master[torch.tensor(alive, dtype=torch.long, device=dev), :n_new] = run.gen[:, :n_new]
            gen = master[:, :n_new].clone()
            if run.Q != Q:                                     # the pools (and the last runner) are those of the survivors only
                self._graphs = {k_: r_ for k_, r_ in self._graphs.items() if r_.kv is not kv}
            for k_ in [k_ for k_ in self._graphs if len(k_) > len(cfgkey) and k_[len(cfgkey)] == "retired"]:
                self._graphs.pop(k_)                           # a grown-only runner: its pools are sized for this call's tail, not the next call's start
        else:
            gen = run.gen[:, :n_new].clone()
        if eos_t is not None:
            gen = self._trim_after_all_finished(gen, eos_t, pad_token_id)
            if scores is not None:
                scores = scores[:gen.shape[1]]
        # prompt ids back on the device in ONE copy (a torch.tensor(..., device=) per question is a synchronous pageable copy each)
        seqs_out = prompt_plus_answer(dev, ids_list, gen)
        stats["steps"] = int(gen.shape[1])
        stats["graph"] = run.graph is not None
        stats["n_groups"] = n_groups          # > 0: the decode steps ran the grouped (shared-prefix) attention
        return GenerateOutput(seqs_out, gen, scores, top_prob, top_tok, stats,
                              attentions=StepAttentions(attn_maps, lm.n_layers, int(gen.shape[1])) if attn_maps is not None else None, branch_top=branch_top)

    # -- a question LIST with a bounded number in flight: waiting questions take the slots of finished ones -----------------------------
    @torch.no_grad()
    def generate_list(self, input_ids: Optional[Sequence[torch.Tensor]], images: Optional[Sequence[torch.Tensor]] = None, in_flight: int = 90,
                      images_cd: Optional[Sequence[torch.Tensor]] = None, inputs_embeds: Optional[Sequence[torch.Tensor]] = None,
                      embeds_prefix: Optional[Sequence[Optional[tuple]]] = None, cd_alpha: Optional[float] = None, cd_beta: Optional[float] = None, use_dd: bool = False, use_dd_unk: bool = False,
                      temperature: Optional[float] = None, top_p: Optional[float] = None, top_k: Optional[int] = None,
                      max_new_tokens: int = 64, eos_token_id=None, pad_token_id: Optional[int] = None, cd_greedy: bool = False,
                      n_top: int = 0, seed: Optional[int] = None, sync_every: int = 8, admit_min: Optional[int] = None) -> GenerateOutput:
        """The reference walks an arbitrarily long question list one generate() call at a time (llava_sampling.py:78-116: LLaVA-Bench, answers
        of 20 - 1,000 tokens; llava_calibrate.py:130).  generate() decodes ONE batch to its end - the last rows of a batch of open-ended
        answers run almost alone.  This call takes the whole list and keeps `in_flight` questions decoding: when questions have emitted EOS
        (or reached max_new_tokens), the next waiting questions are prefilled INTO THEIR SLOTS - own-KV slots, prefix slots and the rows of
        the captured decode step are reused in place, nothing is re-captured or repacked - and join the running batch with a step index of
        their own.  Same kwargs and semantics per question as generate() (LLaVA prompts: ids with one -200 slot + one image each; the
        image-free branches use_dd / use_dd_unk; images_cd = one noised image per question: the VCD branch, which - like the reference's,
        whose cd pass runs on the main cache from step 1 on (quirk #1) - contrasts step 0 only, so it is prefilled at admission into
        scratch prefix slots and never decodes).  Prompts given as embeddings (`input_ids=None, inputs_embeds=[...]`: the Qwen-VL / InstructBLIP
        call shape, with `embeds_prefix` and embedding-valued `images_cd` as in generate(); the image-free branches of such prompts re-run the
        SAME inputs) are admitted the same way.  Processors, output_scores and streamers stay with generate().
        Memory: nb x in_flight own slots of (longest suffix + max_new_tokens) tokens are held for the whole call.
        admit_min: waiting questions are admitted once that many slots are free (default in_flight / 16; prefilling a handful of questions
        costs a pass over the weights like a decode step of the whole batch).  Returns a GenerateOutput over ALL questions, input order;
        stats: admissions, decode steps, mean live rows per step.  In batch-invariant mode with cd_greedy every answer equals the one
        generate() gives the question in any batch."""
        dev, lm = self.device, self.cfg.lm
        if eos_token_id is None:
            raise ValueError("generate_list needs eos_token_id: without an EOS every answer has max_new_tokens tokens and generate() in batches does the same work")
        if pad_token_id is None:
            raise ValueError("If `eos_token_id` is defined, make sure that `pad_token_id` is defined.")   # vcd_sample.py:258-259
        if lm.head_dim != 128:
            raise ValueError("generate_list: head_dim 128 models")
        eos_token_id = [eos_token_id] if isinstance(eos_token_id, int) else list(eos_token_id)
        emb_mode = inputs_embeds is not None
        if emb_mode:
            if input_ids is not None or images is not None:
                raise ValueError("generate_list: inputs_embeds replace input_ids / images")
            emb_all = [e.reshape(-1, lm.d).to(dev, self.dtype) for e in inputs_embeds]
            N = len(emb_all)
            if N == 0 or (embeds_prefix is not None and len(embeds_prefix) != N):
                raise ValueError("generate_list: one (key, n_rows) per prompt in embeds_prefix")
            ids_all, s_img = [[] for _ in range(N)], [0] * N
        else:
            ids_all = [r.reshape(-1).tolist() for r in input_ids]
            N = len(ids_all)
            if N == 0 or images is None or len(images) != N:
                raise ValueError("generate_list: one image per question")
            s_img = []
        for q_, r in enumerate(ids_all if not emb_mode else []):
            if r.count(IMAGE_TOKEN_INDEX) != 1 or r[-1] == IMAGE_TOKEN_INDEX:
                raise ValueError(f"prompt {q_}: exactly one image placeholder (-200), not at the end")
            bad = [t_ for t_ in r if t_ != IMAGE_TOKEN_INDEX and not (0 <= t_ < lm.vocab)]
            if bad:
                raise ValueError(f"prompt {q_}: token id {bad[0]} outside [0, {lm.vocab})")
            s_img.append(r.index(IMAGE_TOKEN_INDEX))
        vcd = images_cd is not None
        if vcd and len(images_cd) != N:
            raise ValueError("generate_list: one images_cd entry per question")
        if vcd:                                                  # vcd_sample.py:148-150: the cd branch takes the place of the first image-free branch
            names = ["main"] + (["none"] if (use_dd and use_dd_unk) else [])
        else:
            names = ["main"] + (["unk"] if use_dd_unk else (["none"] if use_dd else [])) + (["none"] if (use_dd and use_dd_unk) else [])
        nb, contrast = len(names), vcd or len(names) > 1
        alpha = cd_alpha if cd_alpha is not None else 0.5
        beta = cd_beta if cd_beta is not None else 0.1
        warp = WarpSpec(temperature=temperature, top_k=top_k, top_p=top_p)
        n_img_tok = self.cfg.vision.n_patches
        Qc = max(1, min(int(in_flight), N))
        admit_min = max(1, Qc // 16) if admit_min is None else max(1, int(admit_min))
        if emb_mode:
            # a prompt = [shared rows | own rows]: the rows the caller declares common to every prompt with one key (embeds_prefix: '<img>' + the
            # image slots), else - with image-free branches, which re-run the main branch's inputs - everything but the last position, shared by
            # the branches of the question (engine._plan's rule for one generate() call)
            emb_pre = []
            for i, e in enumerate(emb_all):
                T_ = int(e.shape[0])
                ep = embeds_prefix[i] if embeds_prefix is not None else None
                if ep is not None and 0 < int(ep[1]) < T_:
                    emb_pre.append((("emb", ep[0], int(ep[1])), int(ep[1])))
                elif nb > 1 and T_ > 1:
                    emb_pre.append((("q", i), T_ - 1))
                else:
                    emb_pre.append((None, 0))
            suffix_cap = max(int(e.shape[0]) - p_[1] for e, p_ in zip(emb_all, emb_pre))
            t_pre = max([p_[1] for p_ in emb_pre] + [1])
            longest = max(int(e.shape[0]) for e in emb_all)
            shared_keys = set()
            if vcd:
                emb_cd_all = [e.reshape(-1, lm.d).to(dev, self.dtype) for e in images_cd]
                longest = max([longest] + [int(e.shape[0]) for e in emb_cd_all])
        else:
            suffix_cap = max(len(r) - si - 1 for r, si in zip(ids_all, s_img))
            t_pre = max(si + n_img_tok for si in s_img)
            longest = t_pre + suffix_cap
            shared_keys = {(nm, tuple(r[:si])) for r, si in zip(ids_all, s_img) for nm in names[1:]}
        if longest + max_new_tokens > lm.max_pos:
            raise ValueError(f"prompt + max_new_tokens exceed the rotary table (max_pos = {lm.max_pos})")
        asked = Qc
        # VCD: every admitted question's cd prompt (tokens | noised patches | suffix) is prefilled as ONE sequence into a scratch prefix slot
        t_pool = max(t_pre, longest) if vcd else t_pre
        pre_slots = lambda q_: (2 if vcd else 1) * q_ + len(shared_keys) + 1
        Qc = self._fit_in_flight(Qc, lambda q_: (pre_slots(q_), t_pool, nb * q_, suffix_cap + max_new_tokens))
        admit_min = min(admit_min, max(1, Qc // 2))
        n_pre = pre_slots(Qc)
        kv = self.kv(n_pre, t_pool, nb * Qc, suffix_cap + max_new_tokens, frag_only=False)
        if self._prefill_cache is not None and self._prefill_cache["kv"] is kv:
            self._prefill_cache = None                           # (generate(reuse_prefill=True): its kept state lives in these pools)
        eos_t = h2d_long(dev, eos_token_id)
        from .sampling import fresh_offset
        sd = (fresh_offset() if seed is None else int(seed)) & 0x3FFFFFFFFF
        cfgkey = ("list", Qc, nb, max_new_tokens, alpha, beta, warp.t, warp.k, warp.p, use_dd, use_dd_unk, vcd, cd_greedy, tuple(eos_token_id), pad_token_id,
                  ops.GEMM_BATCH_INVARIANT)
        tail = dict(alpha=alpha, beta=beta, warp=warp, contrast=contrast, is_vcd=vcd, both=(use_dd and use_dd_unk), greedy=cd_greedy, eos_t=eos_t,
                    pad=pad_token_id, output_scores=False, n_groups=0, n_items=0, cpi=1, proc={}, admit=True)
        run = self._runner(cfgkey, Qc, nb, max_new_tokens, tail, kv)
        run.reset(sd << 24)
        run.unfinished.zero_()                                   # every slot starts free
        run.slot.copy_(torch.arange(nb * Qc, dtype=torch.int32, device=dev))
        # a free slot decodes a harmless row until a question takes it: one own token at position 0 of its (zeroed) slot
        run.pos.zero_(); run.cpos.zero_(); run.s0.zero_(); run.gen.fill_(pad_token_id)
        run.rows.copy_(torch.tensor([[r_, 1, 0, 0] for r_ in range(nb * Qc)], dtype=torch.int32).to(dev))
        for i in range(lm.n_layers):
            kv.ko[i][:, :, :1].zero_(); kv.vo[i][:, :, :1].zero_()
        master = torch.full((N, max_new_tokens), pad_token_id, dtype=torch.long, device=dev)
        top_prob = torch.zeros(N, n_top, dtype=torch.float32, device=dev) if n_top else None
        top_tok = torch.zeros(N, n_top, dtype=torch.long, device=dev) if n_top else None
        free_pre = list(range(n_pre - 1, -1, -1))
        table: Dict[tuple, dict] = {}                            # prefix key -> {slot, ref, T, feat (kept alive: its address is part of nobody's key)}
        slot_q = [-1] * Qc                                       # question in each slot
        slot_keys: List[list] = [[] for _ in range(Qc)]
        feat_of: Dict[int, torch.Tensor] = {}
        waiting = list(range(N))
        stats = {"n_rows": nb * Qc, "questions": N, "in_flight": Qc, "in_flight_asked": asked, "admissions": 0, "prefill_tokens": 0, "steps": 0, "graph": False}
        live_row_steps = 0

        def retire(slots):
            """finished slots: their tokens go to `master`, their prefixes lose a reference"""
            if not slots:
                return
            idx = torch.tensor(slots, dtype=torch.long, device=dev)
            master.index_copy_(0, torch.tensor([slot_q[q] for q in slots], dtype=torch.long, device=dev), run.gen[idx, :max_new_tokens])
            for q in slots:
                for key in slot_keys[q]:
                    e = table[key]
                    e["ref"] -= 1
                    if e["ref"] == 0:
                        free_pre.append(e["slot"])
                        del table[key]
                im = None if emb_mode else id(images[slot_q[q]])
                slot_q[q], slot_keys[q] = -1, []
                if im is not None and not any(k_[0] == "img" and k_[2] == im for k_ in table):
                    feat_of.pop(im, None)

        def admit(slots):
            """the next len(slots) waiting questions: vision tower for images not seen yet, prefill of new prefixes + all suffixes, first token"""
            qs = [waiting.pop(0) for _ in slots]
            todo = [] if emb_mode else [i for i in qs if id(images[i]) not in feat_of]
            todo = list({id(images[i]): i for i in todo}.values())
            if todo:
                for i, f in zip(todo, self.image_features([images[i] for i in todo])):
                    feat_of[id(images[i])] = f
            new_pre, suffix, dec = [], [], []
            for b, nm in enumerate(names if emb_mode else []):
                for q, i in zip(slots, qs):
                    e_, (key, P) = emb_all[i], emb_pre[i]
                    T_, row, pslot = int(e_.shape[0]), b * Qc + q, 0
                    if key is not None:
                        e = table.get(key)
                        if e is None:
                            e = table[key] = dict(slot=free_pre.pop(), ref=0, T=P)
                            new_pre.append(dict(slot=e["slot"], tokens=[], img=e_[:P], T=P, pos0=0, pslot=0, plen=0))
                        e["ref"] += 1
                        slot_keys[q].append(key)
                        pslot = e["slot"]
                    suffix.append(dict(slot=row, tokens=None, pre=[], img=e_[P:], suf=[], T=T_ - P, pos0=P, pslot=pslot, plen=P))
                    dec.append((T_, T_ - P, [row, T_ + 1, pslot, P]))
            for b, nm in enumerate([] if emb_mode else names):
                for q, i in zip(slots, qs):
                    r, si = ids_all[i], s_img[i]
                    if nm == "main":
                        key, toks, img, T = ("img", tuple(r[:si]), id(images[i])), r[:si], feat_of[id(images[i])], si + n_img_tok
                    elif nm == "unk":
                        key, toks, img, T = ("unk", tuple(r[:si])), r[:si] + [0], None, si + 1       # the <unk> token stays in the prefix (quirk #3)
                    else:
                        key, toks, img, T = ("none", tuple(r[:si])), r[:si], None, si
                    e = table.get(key)
                    if e is None:
                        e = table[key] = dict(slot=free_pre.pop(), ref=0, T=T)
                        new_pre.append(dict(slot=e["slot"], tokens=toks, img=img, T=T, pos0=0, pslot=0, plen=0))
                    e["ref"] += 1
                    slot_keys[q].append(key)
                    suf = r[si + 1:]
                    row = b * Qc + q
                    suffix.append(dict(slot=row, tokens=suf, img=None, T=len(suf), pos0=T, pslot=e["slot"], plen=T))
                    dec.append((T + len(suf), len(suf), [row, T + len(suf) + 1, e["slot"], T]))
            for q, i in zip(slots, qs):
                slot_q[q] = i
            passes = []
            if new_pre:
                x, pos, cpos, slot, seqs, max_tq = self._pack(new_pre)
                passes.append(dict(x=x, pos=pos, cpos=cpos, slot=slot, seqs=seqs, n_seq=len(new_pre), max_tq=max_tq, to_prefix_pool=True, keep_q=False))
            scratch = []
            if vcd:
                # fresh noise per question (llava_sampling.py:88-91): no feature cache, but whole tower chunks
                f_cd = []
                for i0 in range(0, len(qs) if not emb_mode else 0, self.VIT_CHUNK):
                    f_cd += list(self.vit(torch.stack([images_cd[i].reshape(images_cd[i].shape[-3:]).to(dev) for i in qs[i0:i0 + self.VIT_CHUNK]])))
                scratch = [free_pre.pop() for _ in qs]
                if emb_mode:          # images_cd = the noisy-image EMBEDDINGS of the whole prompt (modeling_llama.py:778-782, modeling_qwen.py:1089-1118)
                    cd_seq = [dict(slot=sl, tokens=[], img=emb_cd_all[i], T=int(emb_cd_all[i].shape[0]), pos0=0, pslot=0, plen=0) for sl, i in zip(scratch, qs)]
                else:
                    cd_seq = [dict(slot=sl, tokens=ids_all[i][:s_img[i]], img=f, suf=ids_all[i][s_img[i] + 1:], T=len(ids_all[i]) - 1 + n_img_tok, pos0=0,
                                   pslot=0, plen=0) for sl, i, f in zip(scratch, qs, f_cd)]
                x, pos, cpos, slot, seqs, max_tq = self._pack(cd_seq)
                last, last_seqs = h2d_int32(dev, [s_["q_row0"] + s_["T"] - 1 for s_ in cd_seq], [[j, 1, s_["T"] - 1, s_["slot"], 0, 0] for j, s_ in enumerate(cd_seq)])
                passes.append(dict(x=x, pos=pos, cpos=cpos, slot=slot, seqs=seqs, n_seq=len(cd_seq), max_tq=max_tq, to_prefix_pool=True, last_rows=last.long(),
                                   last_seqs=last_seqs, keep_q=False))
            x, pos, cpos, slot, seqs, max_tq = self._pack(suffix)
            packs_h = ops.flash_packs([[0, 0, 0, 0, s_["pslot"], s_["plen"]] for s_ in suffix]) if (max_tq <= 32 and (not ops.GEMM_BATCH_INVARIANT or ops.FLASH_PACKS_IN_INVARIANT_MODE)) else None
            last, last_seqs, packs = h2d_int32(dev, [s_["q_row0"] + s_["T"] - 1 for s_ in suffix],
                                               [[j, 1, s_["pos0"] + s_["T"] - 1, s_["slot"], s_["pslot"], s_["plen"]] for j, s_ in enumerate(suffix)],
                                               packs_h if packs_h is not None else [[0, -1, -1, -1]])
            passes.append(dict(x=x, pos=pos, cpos=cpos, slot=slot, seqs=seqs, n_seq=len(suffix), max_tq=max_tq, to_prefix_pool=False, last_rows=last.long(),
                               last_seqs=last_seqs, packs=packs if packs_h is not None else None, keep_q=False))
            outs = self.lm.prefill(passes, kv)
            resid, delta = outs[-1]
            logits0 = self.lm.logits(resid, delta)                                  # [nb * k, V], branch-major
            k = len(slots)
            v0 = logits0[:k]
            if vcd:
                c0 = self.lm.logits(*outs[-2])                                      # the cd prompts' last positions; their K/V are not needed again
                d0 = logits0[k:2 * k] if nb == 2 else None
                free_pre.extend(scratch)
                stats["prefill_tokens"] += sum(s_["T"] for s_ in cd_seq)
            else:
                c0 = logits0[k:2 * k] if contrast else None
                d0 = logits0[2 * k:3 * k] if nb == 3 else None
            unf = torch.ones(k, dtype=torch.long, device=dev)
            # the wave draws from its own Philox stream (seed = admission number): row j of a wave and row j of the running batch never
            # see the same (counter, row) pair
            r0 = contrast_sample(v0, c0, d0, alpha=alpha, beta=beta, warp=warp, pick_argmax=cd_greedy, seed=stats["admissions"] + 1, offset=0,
                                 offset_ptr=run.ctr, n_top=n_top, eos_ids=eos_t, pad_id=pad_token_id, unfinished=unf)
            q_idx = torch.tensor(slots, dtype=torch.long, device=dev)
            run.admit(q_idx, r0.tokens, unf, r0.status, [d_[0] for d_ in dec], [d_[1] for d_ in dec], [d_[2] for d_ in dec], pad_token_id)
            if n_top:
                o_idx = torch.tensor(qs, dtype=torch.long, device=dev)
                top_prob.index_copy_(0, o_idx, r0.top_prob); top_tok.index_copy_(0, o_idx, r0.top_tok)
            stats["admissions"] += 1
            stats["prefill_tokens"] += sum(p_["T"] for p_ in new_pre) + sum(s_["T"] for s_ in suffix)

        import time as _time
        ev_pairs, host_admit = [], 0.0

        def timed_admit(slots):
            nonlocal host_admit
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(dev))
            t0 = _time.perf_counter()
            admit(slots)
            host_admit += _time.perf_counter() - t0
            e1.record(torch.cuda.current_stream(dev))
            ev_pairs.append((e0, e1))

        timed_admit(list(range(Qc)))
        steps, marks = 0, []
        while True:
            run.step(kv)
            steps += 1
            if steps % self.LAUNCH_WINDOW == 0:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                marks.append(ev)
                if len(marks) > 1:
                    marks.pop(0).synchronize()
            if steps % sync_every and not (steps in (1, 3) and sync_every > 2):
                continue
            state = torch.cat([run.unfinished, (run.status | run.status0).ne(0).any().long()[None]]).tolist()    # ONE sync
            unf_h, bad = state[:-1], state[-1]
            live_row_steps += nb * sum(unf_h) * min(sync_every, steps)
            if bad:
                break
            Qr = run.Q
            done = [q for q in range(Qr) if not unf_h[q] and slot_q[q] >= 0]
            if not waiting:
                if not any(unf_h):
                    break
                live = [q for q in range(Qr) if unf_h[q]]
                if len(live) <= self.retire_fraction * Qr and Qr > 8:
                    # the tail of the list: nobody is waiting for the finished slots - a smaller step over the survivors (their KV stays where
                    # it is; one capture per halving)
                    retire(done)
                    small = self._runner(cfgkey + ("tail", len(live)), len(live), nb, max_new_tokens, tail, kv)
                    small.adopt_slots(run, torch.tensor(live, dtype=torch.long, device=dev))
                    slot_q, slot_keys = [slot_q[q] for q in live], [slot_keys[q] for q in live]
                    run = small
                    stats["tail_shrinks"] = stats.get("tail_shrinks", 0) + 1
                continue
            free = [q for q in range(Qr) if not unf_h[q]]
            if len(free) >= min(admit_min, len(waiting)) or not any(unf_h):
                retire(done)
                timed_admit(free[:len(waiting)])
        if bool((run.status | run.status0).ne(0).any().item()):
            raise RuntimeError("probability tensor contains either `inf`, `nan` or element < 0")   # torch.multinomial, vcd_sample.py:202
        retire([q for q in range(run.Q) if slot_q[q] >= 0])
        stats.update(admit_gpu_s=round(sum(a_.elapsed_time(b_) for a_, b_ in ev_pairs) / 1e3, 3), admit_host_s=round(host_admit, 3))
        is_eos = (master[:, :, None] == eos_t[None, None, :]).any(-1)
        n_tok = torch.where(is_eos.any(1), is_eos.float().argmax(1) + 1, torch.full((N,), max_new_tokens, device=dev))
        gen = master[:, : int(n_tok.max().item())].clone()
        seqs_out = prompt_plus_answer(dev, ids_all, gen)        # (embedding prompts: ids_all is empty per question - the new tokens alone)
        stats.update(steps=steps, graph=run.graph is not None, answer_tokens=int(n_tok.sum().item()),
                     mean_live_rows=round(live_row_steps / max(steps, 1), 1), n_groups=0)
        return GenerateOutput(seqs_out, gen, None, top_prob, top_tok, stats)

    share_repeated_rows = True        # _plan: image-free rows with identical ids share all but their last position
    REPEATED_ROWS_MIN_SAVED = 256     # ... when that saves at least this many prefill tokens in a branch (every distinct row takes a prefix slot)
    KV_MEMORY_FRACTION = 0.92     # of what is free (+ what the pools being replaced give back): the rest is prefill activations, logits, graphs

    def _fit_in_flight(self, Qc: int, dims) -> int:
        """generate_list holds its own-KV slots at full length (suffix + max_new_tokens) for the whole call: 90 questions x 3 branches x 1,024
        new tokens of a 13B model are 228 GB.  The number in flight is lowered (by eighths, not below 8) until the pools `dims(Qc)` =
        (n_pre, t_pre, n_own, t_own) fit the device; the caller sees the value used in stats['in_flight']."""
        if self.device.type != "cuda":
            return Qc
        cur = self._kvs.get(False)
        free, _total = torch.cuda.mem_get_info(self.device)
        avail = free + torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device) + (cur.nbytes() if cur is not None else 0)

        def need(q_):
            n_pre, t_pre, n_own, t_own = dims(q_)
            t_pre, t_own = (max(t_pre, 64) + 63) // 64 * 64, (max(t_own, 64) + 15) // 16 * 16
            if cur is not None:
                n_pre, t_pre, n_own, t_own = max(n_pre, cur.n_pre), max(t_pre, cur.t_pre), max(n_own, cur.n_own), max(t_own, cur.t_own)
            return KVCache.bytes_needed(self.cfg.lm, n_pre, t_pre, n_own, t_own, self.dtype, False)
        def fits_current(q_):
            n_pre, t_pre, n_own, t_own = dims(q_)
            return cur is not None and cur.fits(n_pre, (max(t_pre, 64) + 63) // 64 * 64, n_own, (max(t_own, 64) + 15) // 16 * 16, False)
        while Qc > 8 and not fits_current(Qc) and need(Qc) > self.KV_MEMORY_FRACTION * avail:
            Qc = max(8, Qc * 7 // 8)
        return Qc

    def _processor_config(self, prompt_lens, ids_list, eos_token_id, min_new_tokens, min_length, stop_words_ids, repetition_penalty,
                          logits_processor, max_new_tokens):
        """-> (proc dict for the runner (empty: no processor stage), hashable key).  Activation rules are HF's
        `_get_logits_processor` [ext]: repetition_penalty iff not None and != 1.0; min_length iff > 0 and an EOS id is defined;
        min_new_tokens iff > 0 and an EOS id is defined."""
        proc, key = {}, []
        eos_defined = eos_token_id is not None and len(eos_token_id) > 0
        floor_on = eos_defined and ((min_new_tokens or 0) > 0 or ((min_length or 0) > 0 and any((min_length - n) > 0 for n in prompt_lens)))
        if floor_on:
            proc["eos_min"] = True
            key.append("eos_min")
        if stop_words_ids is not None:
            if not eos_defined:
                raise ValueError("stop_words_ids needs eos_token_id (the id the stop words force)")
            proc["stop"] = ops.StopWords(stop_words_ids, eos_token_id[0], self.device)
            key.append(("stop", tuple(tuple(w) for w in proc["stop"].seqs), proc["stop"].eos_token_id))
        if repetition_penalty is not None and float(repetition_penalty) != 1.0:
            if not float(repetition_penalty) > 0:
                raise ValueError(f"`penalty` has to be a strictly positive float, but is {repetition_penalty}")
            if any(t == IMAGE_TOKEN_INDEX for r in ids_list for t in r):
                raise ValueError("repetition_penalty with an image placeholder (-200) in input_ids: HF's processor gathers "
                                 "scores[input_ids] and fails on it; the reference only combines it with slot-free prompts")
            proc["rep"] = float(repetition_penalty)
            key.append(("rep", proc["rep"]))
        if logits_processor:
            proc["python"] = list(logits_processor)
            key.append(("python", tuple(id(f) for f in proc["python"])))
        if "rep" in proc or "python" in proc:
            proc["hist_len"] = (max(prompt_lens + [1]) + 63) // 64 * 64
            key.append(proc["hist_len"])
            if proc["hist_len"] + max_new_tokens + 1 > 8192:
                raise ValueError("repetition_penalty / logits_processor: prompt + new tokens exceed the 8192-id history bound")
        return proc, tuple(key)

    def _processor_inputs(self, proc, prompt_lens, ids_list, min_new_tokens, min_length):
        dev, out = self.device, {}
        if proc.get("eos_min"):
            floor = [max(int(min_new_tokens or 0), int(min_length or 0) - n) for n in prompt_lens]
            (out["eos_min"],) = h2d_int32(dev, floor)
        if proc.get("stop") is not None:
            out["prompt_tail"] = proc["stop"].prompt_tail(ids_list, dev)
        if "hist_len" in proc:
            L = max(prompt_lens + [0])
            out["prompt_ids"] = torch.tensor([[-1] * (L - len(r)) + [(-1 if t == IMAGE_TOKEN_INDEX else t) for t in r] for r in ids_list],
                                             dtype=torch.long, device=dev).reshape(len(ids_list), L)
        return out

    def _runner(self, key, Q, nb, max_new, tail, kv):
        r = self._graphs.get(key)
        if r is not None and r.kv is not kv:
            self._graphs.pop(key)
            r = None
        if r is None:
            if len(self._graphs) >= 4:
                self._graphs.pop(next(iter(self._graphs)))
            r = _DecodeRunner(self, Q, nb, max_new, tail, kv)
            self._graphs[key] = r
        return r

    two_level_prefix = True      # [system prompt] prefilled once, the image prefixes behind it (False: every [sys + image] prefix in full)

    @staticmethod
    def _split_system_prompt(plan, min_tokens: int = 8, min_saved: int = 128):
        """Two-level prefixes: the [system prompt + 576 patch embeddings] prefixes of a call all start with the same tokens (the
        conversation template, 35 ids for LLaVA-1.5): that part is prefilled ONCE into a slot of its own, and each image prefix only
        runs its patch rows, attending [parent slot | own rows] and keeping the rows in front for a copy of the parent's K / V - the slot
        the decode steps and the suffix pass read is the same [sys + image] prefix as before, 4 % fewer prefill tokens on the POPE batch.
        Causal attention makes the parent's K / V independent of what follows, so the results are those of the one-level prefill."""
        groups: Dict[tuple, list] = {}
        for s_ in plan["prefix"]:
            if s_.get("img") is not None and s_.get("tokens") and s_["pos0"] == 0:
                groups.setdefault(tuple(s_["tokens"]), []).append(s_)
        # one parent per call keeps the level-1 pass uniform (one row offset); take the prompt shared by the most images
        best = max(groups.items(), key=lambda kv_: len(kv_[1]), default=None)
        # (one question with a VCD branch has two image prefixes: a third pass per layer to save 35 tokens is a loss - the split has to
        #  save at least `min_saved` prefill tokens, i.e. five images behind LLaVA-1.5's 35-token system prompt)
        if best is None or len(best[1]) < 2 or len(best[0]) < min_tokens or len(best[0]) * (len(best[1]) - 1) < min_saved:
            return
        toks, members = list(best[0]), best[1]
        parent = dict(slot=len(plan["prefix"]), tokens=toks, img=None, T=len(toks), pos0=0, pslot=0, plen=0)
        plan["prefix"].append(parent)
        n = len(toks)
        for s_ in members:
            s_.update(full_T=s_["T"], T=s_["T"] - n, tokens=[], pos0=n, cpos0=n, pslot=parent["slot"], plen=n)
        plan["prefill_tokens"] -= n * (len(members) - 1)

    @staticmethod
    def _trim_after_all_finished(gen, eos_t, pad):
        """The reference stops at the first step after which every row has emitted EOS (:291); we check every
        `sync_every` steps, so cut the surplus all-pad columns to return exactly what it returns."""
        is_eos = (gen[:, :, None] == eos_t[None, None, :]).any(-1)
        done_at = torch.where(is_eos.any(1), is_eos.float().argmax(1), torch.full((gen.shape[0],), gen.shape[1] - 1, device=gen.device))
        return gen[:, : int(done_at.max().item()) + 1]

    # -- prefill planning ---------------------------------------------------------------------------------
    def _plan(self, branches, n_img_tok, share_prefix, embeds_only=False, embeds_prefix=None):
        """Every (branch, question) sequence = [prefix | suffix].  Prefix = everything up to and including the
        image slot (main/cd: system prompt + 576 patch embeddings; unk: system prompt + <unk>; none: system prompt);
        identical prefixes (same tokens, same image features) are prefilled ONCE into a prefix slot."""
        prefix_slots: Dict[tuple, dict] = {}
        prefix, suffix = [], []
        n_slots, max_len, unshared = 0, 0, 0
        main_rows = branches[0][1]
        # Prompts given as embeddings: rows fed by the SAME tensor - the Qwen-style degenerate image-free branches, which re-run
        # the main branch's inputs - share everything but the last position as a prefix slot: their prefill K/V
        # are identical by construction, so they are computed once (config #4: half the prefill tokens and KV memory).
        emb_uses: Dict[tuple, int] = {}
        if embeds_only and share_prefix:
            for _, rows_, feats_ in branches:
                for qi in range(len(rows_)):
                    k_ = (feats_[qi].data_ptr(), int(feats_[qi].shape[0]))
                    emb_uses[k_] = emb_uses.get(k_, 0) + 1
        for name, rows, feats in branches:
            if embeds_only:                                  # whole prompt given as embeddings
                for qi in range(len(rows)):
                    T = int(feats[qi].shape[0])
                    key = (feats[qi].data_ptr(), T)
                    unshared += T; max_len = max(max_len, T); n_slots += 1
                    if (share_prefix and embeds_prefix is not None and feats is branches[0][2] and embeds_prefix[qi] is not None
                            and 0 < int(embeds_prefix[qi][1]) < T):
                        # caller-declared common leading rows (same image): one prefix slot per (embedding source, key, length) - the
                        # degenerate branches read the main branch's list, so they land in the same slot.  The promise is about the MAIN
                        # prompts only: a 'cd' branch built from images_cd embeddings carries per-question noise in exactly those rows
                        # (run_qwen.py:182, blip_calibrate.py:80 draw it per question) and never shares them
                        n_pre = int(embeds_prefix[qi][1])
                        pkey = ("emb", id(feats), embeds_prefix[qi][0], n_pre)
                        if pkey not in prefix_slots:
                            prefix_slots[pkey] = dict(slot=len(prefix), tokens=[], img=feats[qi][:n_pre], T=n_pre, pos0=0, pslot=0, plen=0)
                            prefix.append(prefix_slots[pkey])
                        suffix.append(dict(slot=len(suffix), tokens=None, pre=[], img=feats[qi][n_pre:], suf=[], T=T - n_pre, pos0=n_pre,
                                           pslot=prefix_slots[pkey]["slot"], plen=n_pre))
                    elif emb_uses.get(key, 0) >= 2 and T > 1:
                        if key not in prefix_slots:
                            prefix_slots[key] = dict(slot=len(prefix), tokens=[], img=feats[qi][: T - 1], T=T - 1, pos0=0, pslot=0, plen=0)
                            prefix.append(prefix_slots[key])
                        ps = prefix_slots[key]["slot"]
                        suffix.append(dict(slot=len(suffix), tokens=None, pre=[], img=feats[qi][T - 1:], suf=[], T=1, pos0=T - 1, pslot=ps, plen=T - 1))
                    else:
                        suffix.append(dict(slot=len(suffix), tokens=None, pre=[], img=feats[qi], suf=[], T=T, pos0=0, pslot=0, plen=0))
                continue
            # image-free rows with the SAME ids (POPE asks "Is there a <object> in the image?" about hundreds of images: a 768-question batch holds
            # a few dozen distinct `unk` / `none` rows): everything but the last position is one prefix slot per distinct row - prefilled once,
            # attended by all its rows through the grouped pass; each row keeps its last token and what it generates in its own slot
            repeated = set()
            if share_prefix and self.share_repeated_rows and feats is None:
                seen: Dict[tuple, int] = {}
                for ids in rows:
                    k_ = tuple(ids)
                    seen[k_] = seen.get(k_, 0) + 1
                if sum((c_ - 1) * (len(k_) - 1) for k_, c_ in seen.items() if c_ > 1) >= self.REPEATED_ROWS_MIN_SAVED:
                    repeated = {k_ for k_, c_ in seen.items() if c_ > 1 and len(k_) > 1}
            for qi, ids in enumerate(rows):
                src = main_rows[qi]
                s_img = src.index(IMAGE_TOKEN_INDEX) if IMAGE_TOKEN_INDEX in src else None
                img = feats[qi] if (feats is not None and s_img is not None) else None
                if img is not None:                          # main / cd: [tokens before the slot | 576 patch embeddings]
                    pre_tok, suf_tok, plen = ids[:s_img], ids[s_img + 1:], s_img + n_img_tok
                    key = (tuple(pre_tok), img.data_ptr())
                else:
                    if repeated and tuple(ids) in repeated:
                        cut = len(ids) - 1
                    elif s_img is None:
                        cut = self._common_split(rows) if share_prefix else 0
                    else:
                        cut = s_img + 1 if name == "unk" else s_img     # unk keeps the one <unk> token in the prefix (quirk #3)
                    pre_tok, suf_tok, plen = ids[:cut], ids[cut:], cut
                    key = (tuple(pre_tok), None)
                total = plen + len(suf_tok)
                unshared += total
                max_len = max(max_len, total)
                if len(suf_tok) == 0:                      # keep at least the last token in the suffix: its logits are needed
                    if img is None:
                        pre_tok, suf_tok, plen = pre_tok[:-1], pre_tok[-1:], plen - 1
                        key = (tuple(pre_tok), None)
                    else:
                        raise ValueError("prompt must not end with the image token")
                if share_prefix and plen > 0:
                    if key not in prefix_slots:
                        prefix_slots[key] = dict(slot=len(prefix), tokens=pre_tok, img=img, T=plen, pos0=0, pslot=0, plen=0)
                        prefix.append(prefix_slots[key])
                    ps = prefix_slots[key]["slot"]
                    suffix.append(dict(slot=len(suffix), tokens=suf_tok, img=None, T=len(suf_tok), pos0=plen, pslot=ps, plen=plen))
                else:
                    suffix.append(dict(slot=len(suffix), tokens=pre_tok + suf_tok if img is None else None, pre=pre_tok, img=img,
                                       suf=suf_tok, T=total, pos0=0, pslot=0, plen=0))
                n_slots += 1
        tokens = sum(s["T"] for s in prefix) + sum(s["T"] for s in suffix)
        return dict(prefix=prefix, suffix=suffix, n_slots=n_slots, max_len=max_len, prefill_tokens=tokens, unshared_tokens=unshared)

    def _common_split(self, rows):
        """Text-only prompts (no image slot; the content-free `none` / `unk` prior passes of the calibrate drivers, llava_calibrate.py:46-61):
        the longest prefix common to EVERY prompt of the branch - the conversation template's system prompt - is prefilled once.  Computed
        once per branch (cached on the list object); fewer than 8 common tokens or a single prompt: share nothing."""
        import numpy as np
        memo = getattr(self, "_lcp_memo", None)
        if memo is not None and memo[0] is rows:
            return memo[1]
        cut = 0
        if len(rows) >= 2:
            n = min(len(r) for r in rows)
            if n >= 8:
                a = np.array([r[:n] for r in rows])
                same = (a == a[0]).all(0)
                cut = int(n if same.all() else np.argmin(same))
                if cut < 8:
                    cut = 0
        self._lcp_memo = (rows, cut)
        return cut

    def _pack(self, segs):
        """Builds the packed embedding matrix and the per-token / per-sequence descriptors of a prefill pass (numpy on the
        host: one Python iteration per SEQUENCE, none per token; the descriptors go to the device in one copy)."""
        import numpy as np
        dev, t = self.device, self.w.t
        T = np.fromiter((s["T"] for s in segs), dtype=np.int64, count=len(segs))
        q0 = np.concatenate([[0], np.cumsum(T)[:-1]])
        total = int(T.sum())
        for s, r in zip(segs, q0.tolist()):
            s["q_row0"] = r
        x = torch.empty(total, self.cfg.lm.d, dtype=self.dtype, device=dev)
        id_chunks, row_chunks = [], []
        for s, r in zip(segs, q0.tolist()):
            if s.get("tokens") is not None and s["img"] is None:
                id_chunks.append(s["tokens"]); row_chunks.append(np.arange(r, r + s["T"]))
            else:
                pre = s["tokens"] if s.get("tokens") is not None else s["pre"]
                suf = s.get("suf", [])
                n_img = int(s["img"].shape[0])
                x[r + len(pre): r + len(pre) + n_img] = s["img"]
                if pre:
                    id_chunks.append(pre); row_chunks.append(np.arange(r, r + len(pre)))
                if suf:
                    r2 = r + len(pre) + n_img
                    id_chunks.append(suf); row_chunks.append(np.arange(r2, r2 + len(suf)))
        ids_h = np.concatenate([np.asarray(c, dtype=np.int32) for c in id_chunks]) if id_chunks else np.zeros(0, np.int32)
        rows_h = np.concatenate(row_chunks).astype(np.int32) if row_chunks else np.zeros(0, np.int32)
        within = np.arange(total) - np.repeat(q0, T)                                      # index inside the slot (= pos - plen)
        pos_h = (within + np.repeat(np.fromiter((s["pos0"] for s in segs), dtype=np.int64, count=len(segs)), T)).astype(np.int32)
        # (a two-level image prefix writes behind the rows kept for its system prompt: cpos0 = that many rows)
        cpos_h = (within + np.repeat(np.fromiter((s.get("cpos0", 0) for s in segs), dtype=np.int64, count=len(segs)), T)).astype(np.int32)
        slot_h = np.repeat(np.fromiter((s["slot"] for s in segs), dtype=np.int64, count=len(segs)), T).astype(np.int32)
        seqs_h = np.array([[s["q_row0"], s["T"], s["pos0"], s["slot"], s["pslot"], s["plen"]] for s in segs], dtype=np.int32)
        pos, cpos, slot, seqs, ids_d, rows_d = h2d_int32(dev, *(torch.from_numpy(a) for a in (pos_h, cpos_h, slot_h, seqs_h, ids_h, rows_h)))
        if ids_h.size:
            ops.embed_scatter(ids_d, rows_d, t["embed"], x)
        return x, pos, cpos, slot, seqs, max(s["T"] for s in segs)
