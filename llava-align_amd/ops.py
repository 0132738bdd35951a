"""torch-tensor front-ends of the C-ABI model kernels (include/vdd_hip.h).  torch only
provides device memory and the stream; every op here is a hand-written HIP kernel and
raises if the library is missing (no eager fallback)."""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import Optional

import torch

from . import _lib

_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
_SIGS = {
    "vdd_embed_scatter": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vdd_skinny_swiglu": [_P, _P, _P, _I, _I, _I, _L, _I, _P],
    "vdd_decode_attention_fused": [_P] * 11 + [_I, _I, _I, _I, _L, _I, _L, _I, _F, _I, _P],
    "vdd_decode_attention_fused_split": [_P] * 11 + [_I, _I, _I, _I, _L, _I, _L, _I, _F, _P, _I, _I, _P],
    "vdd_rmsnorm": [_P, _P, _P, _I, _P, _P, _P, _I, _I, _F, _I, _P],
    "vdd_gemm": [_P, _P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _L, _I, _I, _P, _L, _I, _P],
    "vdd_rope_kv_write": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _I, _I, _P],
    "vdd_silu_mul": [_P, _P, _L, _I, _I, _P],
    "vdd_embed": [_P, _P, _P, _I, _I, _I, _I, _P],
    "vdd_skinny_gemm": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _L, _L, _I, _P],
    "vdd_decode_attention": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _I, _L, _I, _I, _F, _I, _P],
    "vdd_prefix_fragments": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vdd_decode_attention_grouped": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _L, _I, _L, _I, _I, _I, _I, _F, _I, _P],
    "vdd_flash_attention": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _L, _I, _L, _I, _F, _I, _I, _P],
    "vdd_flash_attention_packed": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _I, _L, _I, _F, _I, _P],
    "vdd_layernorm": [_P, _P, _P, _P, _I, _I, _F, _I, _P],
    "vdd_attention_probs": [_P, _P, _P, _P, _P, _I, _I, _I, _L, _I, _L, _I, _F, _I, _P],
    "vdd_bias_act": [_P, _P, _P, _L, _I, _I, _I, _P],
    "vdd_add": [_P, _P, _P, _L, _I, _P],
    "vdd_vit_im2col": [_P, _I, _P, _I, _I, _I, _I, _I, _P],
    "vdd_vit_assemble": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vdd_vit_qkv_split": [_P, _P, _P, _P, _I, _I, _I, _I, _L, _I, _I, _I, _P],
    "vdd_skinny_gemm_resid_ss": [_P, _P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _I, _P],
    "vdd_skinny_gemm_normed": [_P, _P, _I, _P, _F, _P, _P, _I, _I, _I, _L, _L, _I, _P],
    "vdd_skinny_swiglu_normed": [_P, _P, _I, _P, _F, _P, _P, _I, _I, _I, _L, _I, _P],
    "vdd_stop_words_match": [_P, _L, _L, _P, _P, _I, _P, _P, _I, _P, _I, _P],
    "vdd_repetition_penalty": [_P, _L, _I, _I, _I, _P, _I, _P, _L, _L, _P, _F, C.c_uint32, _P],
}
_bound = False


def _lib_ready():
    global _bound
    lib = _lib.load_lib()
    if not _bound:
        for name, sig in _SIGS.items():
            if hasattr(lib, name):
                fn = getattr(lib, name)
                fn.argtypes, fn.restype = sig, C.c_int
        _bound = True
    return lib


def _st(t: torch.Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


_MODEL_DT = {torch.bfloat16: _lib.VDD_BF16, torch.float16: _lib.VDD_F16}


def _dt(*ts):
    """The storage type of a model-kernel call (the `dtype` argument of the C ABI): every tensor of the call is a device tensor of ONE
    16-bit float type - bf16 (BASELINE config #2) or fp16 (what the reference's drivers load: builder.py:40)."""
    dt = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda or t.dtype not in _MODEL_DT:
            raise ValueError("model kernels take bf16 or fp16 device tensors")
        if dt is not None and t.dtype != dt:
            raise ValueError(f"model kernels take tensors of one dtype per call (got {dt} and {t.dtype})")
        dt = t.dtype
    if dt is None:
        raise ValueError("model kernels take bf16 or fp16 device tensors")
    return _MODEL_DT[dt]


def rmsnorm(x, w, eps, delta=None, resid_out=None, out=None):
    """h = x (+ delta); resid_out <- h; returns h * rsqrt(mean h^2 + eps) * w.   x: [M, d].
    delta: bf16 [M, d], or fp32 [S, M, d] split-K slabs from skinny_gemm(..., n_split=S, slabs=True)."""
    dt = _dt(x, w, resid_out)
    M, d = x.shape
    out = torch.empty_like(x) if out is None else out
    dptr = sptr = None
    ns = 0
    if delta is not None:
        if delta.dtype == torch.float32:
            if delta.dim() != 3 or delta.shape[1:] != x.shape or not delta.is_contiguous():
                raise ValueError("fp32 delta must be contiguous [n_slabs, M, d]")
            sptr, ns = delta.data_ptr(), delta.shape[0]
        else:
            _dt(delta, x)
            dptr = delta.data_ptr()
    _lib.check(_lib_ready().vdd_rmsnorm(x.data_ptr(), dptr, sptr, ns, w.data_ptr(), out.data_ptr(),
                                        resid_out.data_ptr() if resid_out is not None else None, M, d, eps, dt, _st(x)))
    return out


def rope_kv_write(qkv, pos, slot, cos_sin, k_cache, v_cache, Hq, Hkv, D, q_out=None, cpos=None):
    """qkv [M, (Hq+2Hkv)*D]; pos/slot int32 [M]; caches [n_slots, Hkv, t_max, D]; cpos = index inside the slot
    (default: pos).  Returns rotated q [M, Hq*D]."""
    dt = _dt(qkv, k_cache, v_cache)
    M = qkv.shape[0]
    q_out = torch.empty(M, Hq * D, dtype=qkv.dtype, device=qkv.device) if q_out is None else q_out
    cpos = pos if cpos is None else cpos
    _lib.check(_lib_ready().vdd_rope_kv_write(qkv.data_ptr(), pos.data_ptr(), cpos.data_ptr(), slot.data_ptr(), cos_sin.data_ptr(), q_out.data_ptr(),
                                              k_cache.data_ptr(), v_cache.data_ptr(), M, Hq, Hkv, D, k_cache.stride(0),
                                              k_cache.shape[2], dt, _st(qkv)))
    return q_out


def silu_mul(gate_up, out=None):
    dt = _dt(gate_up)
    M, F2 = gate_up.shape
    out = torch.empty(M, F2 // 2, dtype=gate_up.dtype, device=gate_up.device) if out is None else out
    _lib.check(_lib_ready().vdd_silu_mul(gate_up.data_ptr(), out.data_ptr(), M, F2 // 2, dt, _st(gate_up)))
    return out


def embed(ids, table, out=None):
    dt = _dt(table)
    M, d = ids.numel(), table.shape[1]
    out = torch.empty(M, d, dtype=table.dtype, device=table.device) if out is None else out
    _lib.check(_lib_ready().vdd_embed(ids.data_ptr(), table.data_ptr(), out.data_ptr(), M, d, table.shape[0], dt, _st(table)))
    return out


def embed_scatter(ids, rows, table, out):
    """out[rows[m]] = table[ids[m]] (int32 ids / rows): the text chunks of a packed multimodal prompt, in place."""
    dt = _dt(table, out)
    if ids.dtype != torch.int32 or rows.dtype != torch.int32 or ids.numel() != rows.numel():
        raise ValueError("embed_scatter takes int32 ids and rows of equal length")
    _lib.check(_lib_ready().vdd_embed_scatter(ids.data_ptr(), rows.data_ptr(), table.data_ptr(), out.data_ptr(), ids.numel(), table.shape[1], table.shape[0], dt, _st(table)))
    return out


def skinny_gemm(x, w, resid=None, out=None, n_split=1, slabs=False):
    """x [M<=64, K] @ w[N, K]^T (+ resid [M, N]) -> [M, N]; streams w from HBM exactly once.
    slabs=True: returns the fp32 split-K partials [n_split, M, N] instead (feed them to rmsnorm as `delta`)."""
    N = w.shape[0]
    fn = _lib_ready().vdd_skinny_gemm
    dt = _dt(x, w, resid)
    M, K = x.shape
    if slabs:
        out = torch.empty(n_split, M, N, dtype=torch.float32, device=x.device) if out is None else out
        _lib.check(fn(x.data_ptr(), w.data_ptr(), None, None, out.data_ptr(), n_split, M, N, K, x.stride(0), 0, N, dt, _st(x)))
        return out
    out = torch.empty(M, N, dtype=x.dtype, device=x.device) if out is None else out
    _lib.check(fn(x.data_ptr(), w.data_ptr(), resid.data_ptr() if resid is not None else None,
                  out.data_ptr(), None, 1, M, N, K, x.stride(0), resid.stride(0) if resid is not None else 0,
                  out.stride(0), dt, _st(x)))
    return out


SLAB_NORM_MAX_M = 256   # rows up to which a d-wide projection that would take the MFMA GEMM leaves fp32 split-K slabs for its RMSNorm instead (0: off).
                        # (128 until round 5: decode step at 144 rows 7.69 -> 7.43 ms, 256 rows 9.80 -> 9.65, 160 / 192 rows unchanged; beyond 256 rows the
                        # 64 x 256 tiles alone fill a quarter of the chip and slab_splits() returns 0)


def gemm_slabs(x, w, n_split, out=None, tile=8):
    """fp32 [n_split, M, N] partial products of x @ w^T (split-K over n_split parts of K, one (tile, part) per workgroup, no fix-up):
    for a consumer that adds them - rmsnorm(delta=slabs)."""
    dt = _dt(x, w)
    M, K = x.shape
    N = w.shape[0]
    if K % 128 != 0 or N % 4 != 0 or not 1 <= n_split <= min(255, K // 128):
        raise ValueError(f"gemm_slabs: K % 128, N % 4, 1 <= n_split <= K / 128 (got M={M} N={N} K={K} n_split={n_split})")
    out = torch.empty(n_split, M, N, dtype=torch.float32, device=x.device) if out is None else out
    ws = _gemm_workspace(x.device, M, N)
    _lib.check(_lib_ready().vdd_gemm(x.data_ptr(), w.data_ptr(), out.data_ptr(), None, None, M, N, K, x.stride(0), w.stride(0), N, 0,
                                     EPI_NONE, tile + 16 * 3 + (n_split << 8), ws.data_ptr(), ws.numel(), dt, _st(x)))
    return out


def slab_splits(M, N, K, n_cu=256):
    """K parts for gemm_slabs of an [M, K] x [N, K] product on 64 x 256 tiles: as many as give every CU one (tile, part) - but parts of
    at least two 128-deep K units, and none when the tiles alone fill half the chip."""
    tiles = -(-M // 64) * -(-N // 256)
    s = min(n_cu // tiles, K // 256)
    return s if s >= 4 else 0


def linear_to_norm(x, w):
    """Projection whose only consumer is the next RMSNorm's residual add (attention output / MLP down projection): returns either
    the [M, N] product (weight-streaming kernel or MFMA GEMM) or fp32 split-K slabs [S, M, N] that rmsnorm(delta=...) adds, rounds and
    then adds to the residual stream - the same roundings as a rounded product followed by the add.  Which of the three is MEASURED once
    per (rows, N, K, dtype) on the real operands, consumer norm included (`_pick_form`); what decides (tools/skinny_crossover_probe.py,
    DESIGN_APPENDIX): N = d gives the GEMM 16 - 32 output tiles at a few dozen rows - as a finished product each is cut over 8 - 16
    workgroups and put together by ONE that reads the others' partial tiles in turn, as slabs nobody waits; the weight-streaming kernel
    re-reads X per column block and, when N / 16 column blocks do not divide over the CUs (d = 5120: 320 blocks on 256), runs a second
    round for a quarter of the chip."""
    M, K = x.shape
    N = w.shape[0]
    cands = {}
    if _skinny_serves(M, K):
        cands["skinny"] = lambda wi: skinny_gemm(x, wi)
    if K % 128 == 0 and N % 4 == 0:
        cands["gemm"] = lambda wi: gemm(x, wi)
    s_ = slab_splits(M, N, K) if (M <= 256 and N <= 8192 and K % 256 == 0) else 0
    if s_:
        cands["slabs"] = lambda wi: gemm_slabs(x, wi, s_)
    form = _pick_form("to_norm", M, N, K, x, w, cands, consumer="norm")
    if form == "slabs":
        return gemm_slabs(x, w, s_)
    return skinny_gemm(x, w) if form == "skinny" else gemm(x, w)


NORM_FUSED_MAX_M = 16   # rows up to which the normalise-once projections EXIST (their LDS image of the normalised rows; see norm_fused_rows)


FORCE_FORM = {}             # tests / probes: kind ("linear", "to_norm", "swiglu") -> "skinny" | "gemm" | "slabs" wins whenever it is eligible
FORCE_LAYER_FORM = None     # tests / probes: "fused" | "plain" for every row count the fused layer serves


def norm_fused_pays(M: int, d: int, dtype=torch.bfloat16, time_forms=None) -> bool:
    """Does a decode step of M rows take the norm-fused five-launch layer (RMSNorms inside the projections around them) or the
    seven-launch layer (stand-alone norms, the projections in whatever form `_pick_form` measured)?  Measured once per (rows, d, dtype):
    `time_forms()` -> {"fused": fn(i), "plain": fn(i), "device", "n_rot", "iters"}: the engine passes the WHOLE decode step in either form
    (timed on one layer's projections alone the pick was wrong where it is closest: LLaVA-1.5-13B at 2 - 3 rows, where the launches' gaps
    decide - profiles/r06_layer_form_probe.jsonl); without them - or under capture, or with the tuner off - the shape-generic fallback:
    up to 8 rows.  What the measurement finds on the MI355X (tools/fused_band_probe.py, profiles/r05_*): 7B widths fused
    everywhere up to 16 rows except 9 - 12 (the normalise-once kernels change their block plan at 9 rows); d = 5120 only up to 7 rows
    (above, its d-wide projections are faster as split-K slabs, which need the stand-alone norms).  Never in batch-invariant mode."""
    if GEMM_BATCH_INVARIANT or M > norm_fused_rows(d):
        return False
    if FORCE_LAYER_FORM is not None:
        return FORCE_LAYER_FORM == "fused"
    key = ("layer", M, d, 0, _MODEL_DT[dtype])
    got = _form_choice.get(key)
    if got is None and time_forms is not None and GEMM_AUTOTUNE and not torch.cuda.is_current_stream_capturing():
        tf = time_forms()                                  # built only when a measurement is really due
        if tf:
            got = _pick_timed(key, {k: tf[k] for k in ("fused", "plain")}, tf["device"], n_rot=tf["n_rot"], iters=tf.get("iters", 8))
    if got is None:
        return M <= FALLBACK_FUSED_ROWS
    return got == "fused"


def norm_fused_rows(d: int) -> int:
    """How many rows the normalise-once projections take for a residual stream of width d: the normalised rows live in LDS
    (M x 2 d bytes of the 142 KiB the kernel may use)."""
    return min(NORM_FUSED_MAX_M, (142 * 1024) // (2 * d)) if d % 256 == 0 and d <= 8192 else 0


def linear_resid_ss(x, w, resid, out=None, ss=None):
    """h = bf16(bf16(x w^T) + resid) (the new residual stream) and ss [M, N/16] fp32: per-block partial sums of squares of h's rows,
    for linear_normed / swiglu_linear_normed.  M <= NORM_FUSED_MAX_M."""
    dt = _dt(x, w, resid)
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty(M, N, dtype=x.dtype, device=x.device) if out is None else out
    ss = torch.empty(M, (N + 15) // 16, dtype=torch.float32, device=x.device) if ss is None else ss
    _lib.check(_lib_ready().vdd_skinny_gemm_resid_ss(x.data_ptr(), w.data_ptr(), resid.data_ptr(), out.data_ptr(), ss.data_ptr(), M, N, K,
                                                     x.stride(0), resid.stride(0), out.stride(0), dt, _st(x)))
    return out, ss


def linear_normed(h, ss, ln_w, eps, w, out=None, bias=None):
    """rmsnorm(h; ln_w, eps) @ w^T with the normalisation done as h's fragments load (h, ss from linear_resid_ss)."""
    dt = _dt(h, w, ln_w)
    M, K = h.shape
    N = w.shape[0]
    out = torch.empty(M, N, dtype=h.dtype, device=h.device) if out is None else out
    _lib.check(_lib_ready().vdd_skinny_gemm_normed(h.data_ptr(), ss.data_ptr(), ss.shape[1], ln_w.data_ptr(), eps, w.data_ptr(), out.data_ptr(),
                                                   M, N, K, h.stride(0), out.stride(0), dt, _st(h)))
    return bias_act(out, bias, out=out) if bias is not None else out


def swiglu_linear_normed(h, ss, ln_w, eps, w_gate_up, out=None):
    dt = _dt(h, w_gate_up, ln_w)
    M, K = h.shape
    F = w_gate_up.shape[0] // 2
    out = torch.empty(M, F, dtype=h.dtype, device=h.device) if out is None else out
    _lib.check(_lib_ready().vdd_skinny_swiglu_normed(h.data_ptr(), ss.data_ptr(), ss.shape[1], ln_w.data_ptr(), eps, w_gate_up.data_ptr(),
                                                     out.data_ptr(), M, F, K, h.stride(0), dt, _st(h)))
    return out


# ---- which FORM a projection takes is measured, not tabulated (round 6).  Until round 5 the crossovers between the weight-streaming kernels, the
# MFMA GEMM and its split-K slabs were literals measured on three model shapes (SKINNY_ROWS_MEASURED, SKINNY_DEEP_K_MAX_M, SKINNY_WIDE_MAX_M,
# NORM_FUSED_GAP, UNEVEN_FUSED_MAX_M): any other width silently got LLaVA-1.5-7B's.  Now every (kind, rows, N, K, dtype) is timed once on
# the real operands - like the GEMM's tile / schedule - persisted in the same cache file, and the in-tree gfx950 defaults
# (form_choices_mi355x.json, generated by tools/form_sweep.py for the 7B / 13B / Qwen-VL shapes) are such measurements.
FALLBACK_SKINNY_ROWS = 16   # no measurement available (graph capture in progress, GEMM_AUTOTUNE off): weight-streaming up to one MFMA row tile,
FALLBACK_FUSED_ROWS = 8     # the GEMM above; norm-fused layer up to 8 rows - shape-generic, deliberately not a tuned number
_form_choice = {}           # (kind, rows or 64-row bucket, N, K, dtype) -> "skinny" | "gemm" | "slabs";  ("layer", rows, d, 0, dtype) -> "fused" | "plain"


def _skinny_serves(M, K) -> bool:
    """Can the weight-streaming kernels take M rows of a K-deep product?  Up to 16 rows: 16-column blocks, K % 128; 17 - 64 rows: 32-column
    blocks whose two MFMA column tiles share every X fragment, K % 256."""
    return (M <= 16 and K % 128 == 0) or (M <= 64 and K % 256 == 0)


def skinny_rows(N, K):
    """Rows up to which an [N, K] projection takes the weight-streaming kernels according to what has been measured so far (0 in
    batch-invariant mode: one form).  Introspection for tools and tests; the dispatch itself asks `_pick_form` per row count."""
    if GEMM_BATCH_INVARIANT:
        return 0
    dt = _lib.VDD_BF16
    m = 0
    for M in range(1, 65):
        c = _form_choice.get(_form_key("linear", M, N, K, dt))
        if c is None:
            c = "skinny" if (M <= FALLBACK_SKINNY_ROWS and _skinny_serves(M, K)) else "gemm"
        if c != "skinny":
            break
        m = M
    return m


def _form_key(kind, M, N, K, dt):
    return (kind, M if M <= 64 else -(-M // 64) * 64, N, K, dt)


def _time_thunks(thunks, n_rot, iters=8, reps=2):
    """name -> fn(i): fastest name.  Every launch gets another rotation index i (the callers rotate through copies of the weights, so
    that none is still in the 256-MiB Infinity Cache when its turn comes again - in the decode step weights always come from HBM)."""
    best, best_t = None, float("inf")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    turn = 0
    for name, fn in thunks.items():
        fn(turn % n_rot); turn += 1                       # warm-up: nested tuning (the GEMM's tile, inner forms) happens here, untimed
        fn(turn % n_rot); turn += 1
        t = float("inf")
        for _ in range(reps):
            e0.record()
            for _ in range(iters):
                fn(turn % n_rot); turn += 1
            e1.record()
            e1.synchronize()
            t = min(t, e0.elapsed_time(e1))
        if t < best_t:
            best, best_t = name, t
    return best


def _pick_timed(key, thunks, device, n_rot, iters=8):
    """The measured choice for `key`, from the process table, the persisted tables, or a timing run now (None: cannot measure here)."""
    got = _form_choice.get(key)
    if got is not None:
        return got if got in thunks else None
    if not GEMM_AUTOTUNE:
        return None
    _load_persisted(device)
    got = _form_choice.get(key)
    if got is not None and got in thunks:
        return got
    if torch.cuda.is_current_stream_capturing():
        return None                                        # (not cached: the next eager call measures)
    with _CacheLock():                                     # one rank of a node measures, the others read its pick
        _read_cache_section()
        got = _form_choice.get(key)
        if got is None or got not in thunks:
            got = _form_choice[key] = _time_thunks(thunks, n_rot, iters=iters)
            _store_persisted(("form",) + tuple(key), got)
    return got


def _pick_form(kind, M, N, K, x, w, cands, consumer=None):
    """Which of `cands` (name -> fn(weight)) runs this [M, K] x [N, K]^T product.  One eligible form: that one.  Batch-invariant mode: the
    GEMM.  Else the measured one; a product whose consumer is the next RMSNorm is timed WITH that norm (slabs make the norm add S fp32
    partials: cheaper projection, dearer norm)."""
    if GEMM_BATCH_INVARIANT and "gemm" in cands:
        return "gemm"
    if len(cands) == 1:
        return next(iter(cands))
    if FORCE_FORM.get(kind) in cands:
        return FORCE_FORM[kind]
    dt = _MODEL_DT[x.dtype]
    key = _form_key(kind, M, N, K, dt)
    got = _form_choice.get(key)
    if got is not None and got in cands:
        return got
    fallback = "skinny" if ("skinny" in cands and M <= FALLBACK_SKINNY_ROWS) else ("gemm" if "gemm" in cands else next(iter(cands)))
    if not GEMM_AUTOTUNE or torch.cuda.is_current_stream_capturing():
        return fallback
    n_copies = int(min(24, max(2, -(-640 * 2 ** 20 // (w.numel() * 2)))))
    try:
        copies = [w] + [w.clone() for _ in range(n_copies - 1)]
    except torch.OutOfMemoryError:
        copies = [w]
    if consumer == "norm":
        res, lnw = torch.zeros(M, N, dtype=x.dtype, device=x.device), torch.ones(N, dtype=x.dtype, device=x.device)
        thunks = {n_: (lambda i, f=f: rmsnorm(res, lnw, 1e-5, delta=f(copies[i]))) for n_, f in cands.items()}
    else:
        thunks = {n_: (lambda i, f=f: f(copies[i])) for n_, f in cands.items()}
    got = _pick_timed(key, thunks, x.device, len(copies))
    return got if got is not None else fallback


# ---- row-batched MFMA GEMM (csrc/vdd_gemm.hip): every projection above SKINNY_MAX_M rows
EPI_NONE, EPI_BIAS, EPI_BIAS_QUICK_GELU, EPI_BIAS_GELU, EPI_SWIGLU, EPI_BIAS_RESID = range(6)
GEMM_TUNE_MAX_M = 4096          # shapes up to here (the decode batch, single images) pick their tile shape / schedule by a
                                # one-off timing run; above, 256 x 256 tiles + the hybrid schedule (tools/gemm_sched.py)
GEMM_CANDIDATES = [(c, s_) for c in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16) for s_ in (0, 1, 2)]     # (macro tile id, schedule)


def gemm_config(tile: int, sched: int = 0) -> int:
    """The `config` word of vdd_gemm: tile id in bits 0-3 and 6-7, schedule in bits 4-5 (tile ids up to 15 are `tile + 16 * sched` as ever)."""
    return (tile & 15) | ((sched & 3) << 4) | ((tile >> 4) << 6)

import os as _os
_SMALL_TILE_ROWS = [int(v) for v in _os.environ.get("VDD_GEMM_SMALL_TILE_ROWS", "64,32").split(",")]     # probes: rows up to which the 64- / 32-row tiles are candidates
_GEMM_EXCLUDE = {int(c) for c in _os.environ.get("VDD_GEMM_EXCLUDE", "").split(",") if c.strip()}      # probes: tile ids the tuner must not pick
GEMM_BATCH_INVARIANT = False    # True = batch-invariant mode: a row's results no longer depend on which other rows share its batch, because
                                # every op then has ONE form with one summation order per output element -
                                #   * projections: the MFMA GEMM's data-parallel schedule at EVERY row count (each element accumulated over K in one fixed
                                #     order whatever the macro tile; stream-K cuts K where the batch shape puts the cut, the weight-streaming kernels split
                                #     K over waves): skinny_rows() = 0, no norm-fused layer, no split-K slabs;
                                #   * decode attention: rope_kv_write + the per-row split-KV kernel (fixed 64-key chunks merged in key order) for every batch -
                                #     no one-launch kernel for few rows, no grouped prefix pass (its chunking follows the batch): fused_attention_rows() = 0,
                                #     engine: no grouping;
                                #   * prefill: unchanged - its planning choices (prefix sharing, two-level prefixes, packs of four short suffixes) move
                                #     WHERE a key is read from, not the order keys are summed in (64-key tiles at global key indices): measured bit-identical.
                                # What it buys: 1-GPU and N-GPU runs of a deterministic decode (cd_greedy / top_k = 1) agree token for token,
                                # so do a batch and any sub-batch, and row retirement is exact.  What it costs: `batch_invariant` on the bench line.
                                # The sharded drivers select it for such runs (pope_driver.resolve_batch_invariant); `with ops.batch_invariant():` scopes it.
_gemm_ws = {}
_gemm_choice = {}


FLASH_PACKS_IN_INVARIANT_MODE = True    # packs of four short suffixes per attention workgroup walk each sequence's key tiles at the same global key
                                        # indices as one sequence per workgroup: bit-identical (tools/invariance_probe.py, profiles/r06_invariance_probe.jsonl,
                                        # tests/test_batch_invariant_gpu.py) - so they stay on in the mode, like two-level prefixes and prefix sharing


@contextlib.contextmanager
def batch_invariant(on: bool = True):
    """Scope of the batch-invariant mode (GEMM_BATCH_INVARIANT above).  Captured decode steps are keyed by the mode, so entering or
    leaving it never replays a graph of the other form."""
    global GEMM_BATCH_INVARIANT
    old, GEMM_BATCH_INVARIANT = GEMM_BATCH_INVARIANT, bool(on)
    try:
        yield
    finally:
        GEMM_BATCH_INVARIANT = old


def _gemm_workspace(device, M, N):
    """Scratch of the persistent GEMM (arrival counters + one fp32 partial tile per workgroup), one per (device, STREAM): two
    GEMMs that may run at the same time (different streams) must not share counters or slabs; launches on one stream are ordered.
    The counters must start at zero and every completed launch leaves them zero.  A graph captured on a stream keeps using that
    stream's buffer on replay."""
    lib = _lib_ready()
    lib.vdd_gemm_workspace_bytes.restype = C.c_int64
    lib.vdd_gemm_workspace_bytes.argtypes = [_I, _I]
    need = lib.vdd_gemm_workspace_bytes(int(M), int(N))
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _gemm_ws.get(key)
    if ws is None or ws.numel() < need:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the GEMM workspace must exist before a graph capture (run the step once eagerly on this stream)")
        ws = _gemm_ws[key] = torch.zeros(need, dtype=torch.uint8, device=device)
    return ws


def gemm_workspace_reset(device=None):
    """Zero the arrival counters of every GEMM workspace (of `device`): needed only after a launch that did not complete."""
    for (dev, _), ws in _gemm_ws.items():
        if device is None or dev == torch.device(device):
            ws[: 4 << 20].zero_()


def _gemm_call(x, w, out, bias, resid, M, N, K, epi, config, ws):
    dt = _MODEL_DT[x.dtype]                                   # (gemm() has checked the operands)
    _lib.check(_lib_ready().vdd_gemm(x.data_ptr(), w.data_ptr(), out.data_ptr(), bias.data_ptr() if bias is not None else None,
                                     resid.data_ptr() if resid is not None else None, M, N, K, x.stride(0), w.stride(0), out.stride(0),
                                     resid.stride(0) if resid is not None else 0, epi, config, ws.data_ptr(), ws.numel(), dt, _st(x)))


def gemm(x, w, bias=None, resid=None, epi=EPI_NONE, out=None, config=None):
    """out[M, N] = epilogue(x[M, K] @ w[N, K]^T) on the hand-written MFMA kernel (bf16 or fp16 operands, fp32 accumulate).  epi=EPI_SWIGLU:
    w = [Wgate; Wup], N = w.shape[0] // 2.  No library fallback: unsupported shapes raise."""
    dt = _dt(x, w, bias, resid)
    M, K = x.shape
    N = w.shape[0] // 2 if epi == EPI_SWIGLU else w.shape[0]
    if K % 128 != 0 or N % 4 != 0 or x.stride(1) != 1 or w.stride(1) != 1:
        raise ValueError(f"vdd_gemm needs K % 128 == 0, N % 4 == 0 and K-contiguous operands (got M={M} N={N} K={K})")
    out = torch.empty(M, N, dtype=x.dtype, device=x.device) if out is None else out
    if M == 0:
        return out
    ws = _gemm_workspace(x.device, M, N)
    if config is None:
        key = _gemm_key(M, N, K, epi, dt)
        config = _gemm_choice.get(key)
        if config is None and GEMM_AUTOTUNE:
            _load_persisted(x.device)
            config = _gemm_choice.get(key)
        if config is None:
            config = gemm_config(1, 1) if GEMM_BATCH_INVARIANT else 1
            if M <= GEMM_TUNE_MAX_M and GEMM_AUTOTUNE:
                if not torch.cuda.is_current_stream_capturing():
                    with _CacheLock():                             # one rank of a node tunes, the others read its pick
                        _read_cache_section()
                        config = _gemm_choice.get(key)
                        if config is None:
                            config = _gemm_choice[key] = _gemm_tune(x, w, out, bias, resid, M, N, K, epi, ws)
                            _store_persisted(key, config)
                # (under capture: the default for this launch only - caching it would pin an untuned choice for the process)
            else:
                _gemm_choice[key] = config
    _gemm_call(x, w, out, bias, resid, M, N, K, epi, config, ws)
    return out


GEMM_AUTOTUNE = True            # False: 256 x 256 tiles + the hybrid schedule for every shape (no timing runs at all)


def _gemm_key(M, N, K, epi, dt=_lib.VDD_BF16):
    """Tuning granularity: shapes up to GEMM_TUNE_MAX_M rows are bucketed by their number of 64-row units (the decode batch and
    the per-image ViT calls repeat a handful of sizes; a prefill length that differs by a few tokens must not re-run 24 candidates
    and clone 640 MiB of weights); everything above shares one entry.  Up to 32 rows: a bucket of its own (-1) - the 32 x 128 tiles
    (round 6) only serve those."""
    return (-1 if M <= 32 else (-(-M // 64) if M <= GEMM_TUNE_MAX_M else 0), N, K, epi, GEMM_BATCH_INVARIANT, dt)


def gemm_choices_export() -> dict:
    """The tuner's choices so far as a JSON-able dict: "bucket,N,K,epi,batch_invariant,dtype" -> GEMM config, and
    "form,kind,rows,N,K,dtype" -> the measured projection / layer form."""
    out = {",".join(map(str, k)): v for k, v in _gemm_choice.items()}
    out.update({"form," + ",".join(map(str, k)): v for k, v in _form_choice.items()})
    return out


def form_choices_export() -> dict:
    return {"form," + ",".join(map(str, k)): v for k, v in _form_choice.items()}


def gemm_choices_import(d: dict, keep_existing: bool = False):
    for k, v in d.items():
        if k.startswith("form,"):
            _f, kind, M, N, K, dt = k.split(",")
            key = (kind, int(M), int(N), int(K), int(dt))
            if not (keep_existing and key in _form_choice):
                _form_choice[key] = str(v)
            continue
        b, N, K, epi, inv, *dt = k.split(",")
        key = (int(b), int(N), int(K), int(epi), inv == "True", int(dt[0]) if dt else _lib.VDD_BF16)
        if not (keep_existing and key in _gemm_choice):
            _gemm_choice[key] = int(v)


# ---- persistence of the tuner's choices: run-to-run identical logits.  The winner among (tile, schedule) candidates is picked by
# wall-clock timing and a stream-K cut changes the fp32 summation order of a tile, so two processes that tune for themselves may
# settle on different (equally valid) low-order bits - and then sample different tokens from the same seed, where the reference is
# run-to-run deterministic on one machine.  So choices are looked up, in this order: the process's own table; the user's cache file
# (`VDD_GEMM_CHOICES=<path>`, default ~/.cache/llava_align_amd/gemm_choices.json; "off" disables persistence) under the section of
# this device and this build of the library; the in-tree defaults measured on MI355X (gemm_choices_mi355x.json) for the shapes of the
# supported models (`VDD_GEMM_DEFAULTS=off` skips them: re-tuning after a kernel change).  Only a shape found nowhere is timed, once per machine: the result is written back to the cache file at once.
_persist = {"loaded": None, "path": None, "section": None}


def _choices_file():
    import os
    v = os.environ.get("VDD_GEMM_CHOICES")
    if v is not None and v.lower() in ("off", "0", "none", ""):
        return None
    return v if v else os.path.join(os.path.expanduser("~"), ".cache", "llava_align_amd", "gemm_choices.json")


def _lib_fingerprint() -> str:
    import hashlib
    h = hashlib.sha256()
    with open(_lib.lib_path(), "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]


def gemm_source_fingerprint() -> str:
    """What the in-tree defaults are bound to: the GEMM kernel's SOURCE (csrc/vdd_gemm.hip + csrc/vdd_elem.h), not the binary - two
    builds of the same source differ in embedded paths.  "" when the sources do not travel with the package."""
    import hashlib
    import os
    h = hashlib.sha256()
    for f in ("vdd_gemm.hip", "vdd_elem.h"):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", f)
        if not os.path.exists(path):
            return ""
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def kernel_source_fingerprint() -> str:
    """What the in-tree FORM defaults are bound to: the sources of every kernel the forms choose between (the GEMM and the weight-streaming /
    norm kernels)."""
    import hashlib
    import os
    h = hashlib.sha256()
    for f in ("vdd_gemm.hip", "vdd_llm_kernels.hip", "vdd_elem.h"):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", f)
        if not os.path.exists(path):
            return ""
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _read_cache_section():
    """The cache file's choices for this device and library build (re-read on every call: another rank may have tuned meanwhile)."""
    import json
    import os
    if _persist["path"] and os.path.exists(_persist["path"]):
        try:
            with open(_persist["path"]) as f:
                gemm_choices_import(json.load(f).get(_persist["section"], {}))
        except (OSError, ValueError):
            pass                                                   # an unreadable cache is no cache


def _load_persisted(device):
    """Once per process (and per device name): the in-tree defaults, then the user's cache file on top."""
    import json
    import os
    name = torch.cuda.get_device_name(device)
    if _persist["loaded"] == name:
        return
    _persist.update(loaded=name, path=_choices_file(), section=f"{name}|{_lib_fingerprint()}")
    default = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_choices_mi355x.json")
    arch = getattr(torch.cuda.get_device_properties(device), "gcnArchName", "")
    if arch.startswith("gfx950") and os.path.exists(default) and os.environ.get("VDD_GEMM_DEFAULTS", "").lower() not in ("off", "0", "none"):
        with open(default) as f:
            doc = json.load(f)
        # defaults measured on another revision of the kernel are still valid configurations, but no longer the measured winners:
        # skip them (the shapes are then tuned once per machine and cached) unless the caller insists (VDD_GEMM_DEFAULTS=force)
        bound, here = doc.get("gemm_source_sha", ""), gemm_source_fingerprint()
        if not bound or not here or bound == here or os.environ.get("VDD_GEMM_DEFAULTS", "").lower() == "force":
            for k, v in doc.get("choices", {}).items():
                kk = k.split(",")
                key = (int(kk[0]), int(kk[1]), int(kk[2]), int(kk[3]), kk[4] == "True", int(kk[5]))
                _gemm_choice.setdefault(key, int(v))
    fdefault = os.path.join(os.path.dirname(os.path.abspath(__file__)), "form_choices_mi355x.json")
    if arch.startswith("gfx950") and os.path.exists(fdefault) and os.environ.get("VDD_GEMM_DEFAULTS", "").lower() not in ("off", "0", "none"):
        with open(fdefault) as f:
            doc = json.load(f)
        bound, here = doc.get("kernel_source_sha", ""), kernel_source_fingerprint()
        if not bound or not here or bound == here or os.environ.get("VDD_GEMM_DEFAULTS", "").lower() == "force":
            gemm_choices_import(doc.get("choices", {}), keep_existing=True)
    _read_cache_section()


class _CacheLock:
    """Exclusive advisory lock beside the cache file: the ranks of one node tune a missing shape ONE AT A TIME - the second one finds
    the first one's pick in the file instead of timing its own (and possibly landing on another, equally fast, summation order).
    Re-entrant within the process: measuring a projection FORM runs the GEMM, whose own tile tuner takes the lock again (flock on a second
    descriptor of the same file would wait for the first one forever)."""
    _depth = 0
    _file = None

    def __enter__(self):
        import os
        cls = _CacheLock
        if cls._depth == 0:
            cls._file = None
            path = _persist["path"]
            if path:
                try:
                    import fcntl
                    os.makedirs(os.path.dirname(path), exist_ok=True)
                    cls._file = open(path + ".lock", "w")
                    fcntl.flock(cls._file, fcntl.LOCK_EX)
                except (OSError, ImportError):
                    cls._file = None
        cls._depth += 1
        return self

    def __exit__(self, *exc):
        cls = _CacheLock
        cls._depth -= 1
        if cls._depth == 0 and cls._file is not None:
            try:
                import fcntl
                fcntl.flock(cls._file, fcntl.LOCK_UN)
            finally:
                cls._file.close()
                cls._file = None
        return False


def _store_persisted(key, cfg):
    """Append one freshly tuned choice to the cache file (read - merge - atomic replace; the caller holds _CacheLock)."""
    import json
    import os
    path = _persist["path"]
    if not path:
        return
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = {}
        if os.path.exists(path):
            with open(path) as f:
                data = json.load(f)
        data.setdefault(_persist["section"], {})[",".join(map(str, key))] = cfg if isinstance(cfg, str) else int(cfg)
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "w") as f:
            json.dump(data, f, indent=0, sort_keys=True)
        os.replace(tmp, path)
    except (OSError, ValueError):
        pass


def _gemm_tune(x, w, out, bias, resid, M, N, K, epi, ws, iters=8):
    """Times every (tile shape, schedule) candidate on the real operands and keeps the fastest (all write the same result).
    In the decode step a projection's weights always come from HBM (13 GB of them stream through per step), so the timed launches
    rotate through enough copies of `w` that none is still in the 256-MiB Infinity Cache when its turn comes again: timed on ONE
    hot copy the tuner preferred small macro tiles that then ran 10 % slower in place."""
    n_copies = int(min(24, max(2, -(-640 * 2 ** 20 // (w.numel() * 2)))))
    try:
        copies = [w] + [w.clone() for _ in range(n_copies - 1)]
    except torch.OutOfMemoryError:
        copies = [w]
    best, best_t = 1, float("inf")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    turn = 0
    # short launches (up to 128 rows: 30 - 60 us) are timed over up to six batches of `iters` launches, twice (the faster run counts):
    # with 8 launches per candidate the pick among near-equal candidates - and with it the step time at 17 - 64 rows - moved by up to
    # 30 % from process to process (tools/step_curve.py).  Decode-batch and prefill shapes keep one batch of 8 (100 us and up per launch).
    flops = 2.0 * M * N * K * (2 if epi == EPI_SWIGLU else 1)
    short = M <= 128
    chunks, reps = (int(min(6, max(1, 2.5e-3 / max(flops / 1.0e15, 30e-6) / iters))), 2) if short else (1, 1)
    for c, sch in GEMM_CANDIDATES:
        if ((epi == EPI_SWIGLU and c in (5, 6, 7, 9, 13, 14, 16)) or (GEMM_BATCH_INVARIANT and sch != 1) or (c in (8, 9, 10, 11) and M > 256) or (c in (12, 13) and M > _SMALL_TILE_ROWS[0])
                or (c in (14, 15) and M > _SMALL_TILE_ROWS[1]) or (c == 16 and M > 576)):
            continue
        if c in _GEMM_EXCLUDE:
            continue
        cfg = gemm_config(c, sch)
        _gemm_call(x, copies[turn % len(copies)], out, bias, resid, M, N, K, epi, cfg, ws); turn += 1
        t = float("inf")
        for _rep in range(reps):
            tt = 0.0
            for _chunk in range(chunks):
                e0.record()
                for _ in range(iters):
                    _gemm_call(x, copies[turn % len(copies)], out, bias, resid, M, N, K, epi, cfg, ws); turn += 1
                e1.record()
                e1.synchronize()
                tt += e0.elapsed_time(e1)
            t = min(t, tt)
        if t < best_t:
            best, best_t = cfg, t
    _gemm_call(x, w, out, bias, resid, M, N, K, epi, best, ws)          # leave the caller's result in `out`
    return best


def linear(x, w, out=None, bias=None):
    """Row-batched projection: a weight-streaming kernel or the MFMA GEMM, whichever was measured faster for this (rows, N, K, dtype)."""
    M, K = x.shape
    N = w.shape[0]
    cands = {}
    if _skinny_serves(M, K):
        cands["skinny"] = lambda wi: skinny_gemm(x, wi)
    if K % 128 == 0 and N % 4 == 0:
        cands["gemm"] = lambda wi: gemm(x, wi)
    if not cands:
        raise ValueError(f"linear: no kernel serves M={M} N={N} K={K} (K % 128 == 0 and N % 4 == 0 for the GEMM)")
    if _pick_form("linear", M, N, K, x, w, cands) == "skinny":
        y = skinny_gemm(x, w, out=out)
        return bias_act(y, bias, out=y) if bias is not None else y
    return gemm(x, w, bias=bias, epi=EPI_BIAS if bias is not None else EPI_NONE, out=out)


def _skinny_swiglu(x, w_gate_up, out=None):
    M, K = x.shape
    F = w_gate_up.shape[0] // 2
    dt = _dt(x, w_gate_up)
    out = torch.empty(M, F, dtype=x.dtype, device=x.device) if out is None else out
    _lib.check(_lib_ready().vdd_skinny_swiglu(x.data_ptr(), w_gate_up.data_ptr(), out.data_ptr(), M, F, K, x.stride(0), dt, _st(x)))
    return out


def swiglu_linear(x, w_gate_up, out=None):
    """silu(x Wg^T) * (x Wu^T) with w_gate_up = [Wg; Wu]: one launch either way - the fused weight-streaming kernel (<= 16 rows: 8 features per
    block; 17 - 64: 16, gate + up tiles) or the MFMA GEMM with the SwiGLU epilogue (no [M, 2F] round trip, no silu_mul launch); measured."""
    M, K = x.shape
    F = w_gate_up.shape[0] // 2
    cands = {}
    if K % 128 == 0 and F % 128 == 0:
        cands["gemm"] = lambda wi, o=None: gemm(x, wi, epi=EPI_SWIGLU, out=o)
    elif K % 128 == 0 and F % 8 == 0:                       # (odd feature counts: the plain GEMM + vdd_silu_mul, whose lanes take 8 features)
        cands["gemm"] = lambda wi, o=None: silu_mul(gemm(x, wi), out=o)
    if _skinny_serves(M, K):
        cands["skinny"] = lambda wi, o=None: _skinny_swiglu(x, wi, o)
    if not cands:
        raise ValueError(f"swiglu_linear: no kernel serves M={M} F={F} K={K}")
    form = _pick_form("swiglu", M, 2 * F, K, x, w_gate_up, {k: (lambda wi, f=f: f(wi)) for k, f in cands.items()})
    return cands[form](w_gate_up, out)


_attn_ws = {}


def decode_attention(q, k_cache, v_cache, rows, H, Hkv, D, out=None, max_len=None, k_prefix=None, v_prefix=None, workspace=None):
    """q [M, H*D]; rows int32 [M, 4] = (slot, len, prefix_slot, prefix_len); max_len bounds every len.
    k_prefix/v_prefix: separate pool holding the shared prefixes (default: the same buffers, index t)."""
    dt = _dt(q, k_cache, v_cache)
    M = q.shape[0]
    lib = _lib_ready()
    k_prefix = k_cache if k_prefix is None else k_prefix
    v_prefix = v_cache if v_prefix is None else v_prefix
    max_len = int(max_len) if max_len is not None else k_cache.shape[2] + k_prefix.shape[2]
    lib.vdd_decode_attention_workspace_bytes.restype = C.c_int64
    need = lib.vdd_decode_attention_workspace_bytes(M, H, D, max_len)
    key = (q.device, )
    if workspace is not None:
        if workspace.numel() * 4 < need:
            raise ValueError("attention workspace too small")
        ws = workspace
    else:
        ws = _attn_ws.get(key)
        if ws is None or ws.numel() * 4 < need:
            ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=q.device)
            _attn_ws[key] = ws
    out = torch.empty_like(q) if out is None else out
    _lib.check(lib.vdd_decode_attention(q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), k_prefix.data_ptr(), v_prefix.data_ptr(),
                                        rows.data_ptr(), out.data_ptr(), ws.data_ptr(), M, H, Hkv, D, k_cache.stride(0), k_cache.shape[2],
                                        k_prefix.stride(0), k_prefix.shape[2], max_len, D ** -0.5, dt, _st(q)))
    return out


FUSED_ATTN_MAX_M = 16     # rows up to which RoPE + KV write + attention + merge run as one launch
# ... and, for rows whose prefixes are not worth grouping anyway (one image per question, 3 branches), up to here: against rope_kv +
# split-KV attention + combine the one launch is worth 7 % of the step at 18 rows, 2 % at 24, nothing at 33 and costs 2.6 % at 48
# (tools/few_row_curve.py, profiles/r05_few_row_curve.jsonl)
FUSED_ATTN_UNGROUPED_MAX_M = 32


def fused_attention_rows() -> int:
    """Rows up to which an UNGROUPED decode step takes the one-launch RoPE + KV write + attention kernel.  Under GEMM_BATCH_INVARIANT: none
    (the split-KV kernel sums a row's keys in one order at every batch size; a batch that shrinks into the one-launch band would change it)."""
    return 0 if GEMM_BATCH_INVARIANT else max(FUSED_ATTN_MAX_M, FUSED_ATTN_UNGROUPED_MAX_M)


FUSED_ATTN_SPLIT = True   # cut the keys of a (row, head) over 2 / 4 workgroups while H x M of them would leave most CUs idle
_fused_split_ws = {}


def fused_attention_split(M, H):
    """Key slices per (row, head) of the one-launch decode attention: 4 up to 64 (row, head) pairs (one question, two branches of 32
    heads: 256 workgroups instead of 64), 2 up to 128, else 1."""
    if not FUSED_ATTN_SPLIT:
        return 1
    return 4 if M * H <= 64 else (2 if M * H <= 128 else 1)


def _fused_split_workspace(device, M, H, n_split):
    """Partials + tickets of the split form, one buffer per (device, stream, shape); zeroed once (the tickets return to zero after
    every launch).  Must exist before a graph capture, like the GEMM workspace."""
    lib = _lib_ready()
    lib.vdd_decode_attention_fused_split_workspace_bytes.restype = C.c_int64
    key = (device, torch.cuda.current_stream(device).cuda_stream, M, H, n_split)
    ws = _fused_split_ws.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the split-attention workspace must exist before a graph capture (run the step once eagerly on this stream)")
        ws = _fused_split_ws[key] = torch.zeros(lib.vdd_decode_attention_fused_split_workspace_bytes(M, H, n_split), dtype=torch.uint8, device=device)
    return ws


def decode_attention_fused(qkv, pos, cpos, slot, cos_sin, k_cache, v_cache, rows, H, Hkv, D, out=None, k_prefix=None, v_prefix=None, n_split=None):
    """Small-M decode attention straight from the un-rotated qkv projection [M, (H+2Hkv)*D]: RoPE, KV-cache write of the new
    token (index cpos of slot), whole-context attention and merge in one kernel.  Same arguments as rope_kv_write +
    decode_attention; rows[:, 1] (len) counts the new token.  n_split (default: fused_attention_split(M, H)) > 1 cuts the old keys of
    every (row, head) over that many workgroups; the last one to finish merges their partials in slice order."""
    dt = _dt(qkv, k_cache, v_cache)
    M = qkv.shape[0]
    k_prefix = k_cache if k_prefix is None else k_prefix
    v_prefix = v_cache if v_prefix is None else v_prefix
    out = torch.empty(M, H * D, dtype=qkv.dtype, device=qkv.device) if out is None else out
    n_split = fused_attention_split(M, H) if n_split is None else n_split
    if n_split > 1:
        ws = _fused_split_workspace(qkv.device, M, H, n_split)
        _lib.check(_lib_ready().vdd_decode_attention_fused_split(qkv.data_ptr(), pos.data_ptr(), cpos.data_ptr(), slot.data_ptr(), cos_sin.data_ptr(),
                                                                k_cache.data_ptr(), v_cache.data_ptr(), k_prefix.data_ptr(), v_prefix.data_ptr(),
                                                                rows.data_ptr(), out.data_ptr(), M, H, Hkv, D, k_cache.stride(0), k_cache.shape[2],
                                                                k_prefix.stride(0), k_prefix.shape[2], D ** -0.5, ws.data_ptr(), n_split, dt, _st(qkv)))
        return out
    _lib.check(_lib_ready().vdd_decode_attention_fused(qkv.data_ptr(), pos.data_ptr(), cpos.data_ptr(), slot.data_ptr(), cos_sin.data_ptr(),
                                                      k_cache.data_ptr(), v_cache.data_ptr(), k_prefix.data_ptr(), v_prefix.data_ptr(),
                                                      rows.data_ptr(), out.data_ptr(), M, H, Hkv, D, k_cache.stride(0), k_cache.shape[2],
                                                      k_prefix.stride(0), k_prefix.shape[2], D ** -0.5, dt, _st(qkv)))
    return out


def attention_workspace(M, H, D, max_len, device):
    lib = _lib_ready()
    lib.vdd_decode_attention_workspace_bytes.restype = C.c_int64
    return torch.empty((lib.vdd_decode_attention_workspace_bytes(M, H, D, int(max_len)) + 3) // 4, dtype=torch.float32, device=device)


def prefix_chunks_per_item(groups, n_heads, target_waves=2048, max_chunks=16):
    """How many 64-key chunks one work item of the MFMA prefix pass should walk: as many as possible (fewer partials for
    the combine to merge; measured 304 -> 274 us per layer at 768 rows, tools/attn_probe.py) while the pass still has
    ~target_waves waves (one question in flight must stay split: 10 chunks x 32 heads is all the parallelism there is)."""
    total = sum(-(-n_rows // 16) * -(-plen // 64) for _, n_rows, _, plen in groups)
    return max(1, min(max_chunks, total * n_heads // target_waves))


def prefix_work_items(groups, chunks_per_item=1):
    """groups [[row_off, n_rows, pslot, plen], ...] -> work list [[group, first_row, item, 0], ...] of the prefix pass; item j
    covers keys [j * 64 * chunks_per_item, (j + 1) * 64 * chunks_per_item) of the group's prefix.  Longest items first: the pass
    is one wave per (item, head) with no other load balancing, so the short ones (the image-free group's) fill the tail."""
    items = []
    keys = 64 * chunks_per_item
    for gi, (_, n_rows, _, plen) in enumerate(groups):
        for r0 in range(0, n_rows, 16):
            for c in range((plen + keys - 1) // keys):
                items.append((-min(keys, plen - c * keys), len(items), [gi, r0, c, 0]))
    return [it for _, _, it in sorted(items)]


def prefix_fragments(k_prefix, v_prefix, prefix_frag, prefix_len_of_slot):
    """k_prefix / v_prefix [n_slots, Hkv, t_max, D] -> prefix_frag [n_slots, Hkv, 2 * t_max, D]: per 64-key chunk one 32-KiB block of
    MFMA operand images (16 K fragments, 16 V^T fragments) for the grouped decode pass."""
    dt = _dt(k_prefix, v_prefix, prefix_frag)
    n, Hkv, t_max, D = k_prefix.shape
    if tuple(prefix_frag.shape) != (n, Hkv, 2 * t_max, D) or v_prefix.shape != k_prefix.shape or not prefix_frag.is_contiguous():
        raise ValueError("prefix_frag must be a contiguous [n_slots, Hkv, 2 * t_max, D] tensor")
    _lib.check(_lib_ready().vdd_prefix_fragments(k_prefix.data_ptr(), v_prefix.data_ptr(), prefix_frag.data_ptr(), prefix_len_of_slot.data_ptr(),
                                                 prefix_len_of_slot.numel(), Hkv, t_max, D, dt, _st(k_prefix)))
    return prefix_frag


def decode_attention_grouped(q, k_cache, v_cache, k_prefix, v_prefix, rows, groups, group_rows, items, n_items,
                             H, Hkv, D, max_prefix_len, max_own_len, out=None, workspace=None, prefix_frag=None, chunks_per_item=1, scale=None):
    """decode_attention with the shared prefixes attended once per group of rows (MFMA over the group's queries)."""
    dt = _dt(q, k_cache, v_cache, k_prefix, v_prefix)
    M = q.shape[0]
    lib = _lib_ready()
    r64 = lambda v: (int(v) + 63) // 64 * 64
    lib.vdd_decode_attention_workspace_bytes.restype = C.c_int64
    need = lib.vdd_decode_attention_workspace_bytes(M, H, D, r64(max_prefix_len) + r64(max_own_len))
    ws = workspace if workspace is not None else _attn_ws.get((q.device,))
    if ws is None or ws.numel() * 4 < need:
        if workspace is not None:
            raise ValueError("attention workspace too small")
        ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=q.device)
        _attn_ws[(q.device,)] = ws
    out = torch.empty_like(q) if out is None else out
    _lib.check(lib.vdd_decode_attention_grouped(q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), k_prefix.data_ptr(), v_prefix.data_ptr(),
                                                prefix_frag.data_ptr() if prefix_frag is not None else None, rows.data_ptr(), groups.data_ptr(), group_rows.data_ptr(), items.data_ptr(), n_items,
                                                out.data_ptr(), ws.data_ptr(), M, H, Hkv, D, k_cache.stride(0), k_cache.shape[2],
                                                k_prefix.stride(0), k_prefix.shape[2], int(max_prefix_len), int(max_own_len),
                                                int(chunks_per_item), D ** -0.5 if scale is None else scale, dt, _st(q)))
    return out


def flash_attention(q, k_cache, v_cache, seqs, n_seq, max_tq, H, Hkv, D, causal=True, out=None, k_prefix=None, v_prefix=None, scale=None,
                    own_row_offset=0):
    """Prefill attention.  q [Ttot, H*D] packed by sequence; seqs int32 [n_seq, 6] =
    (q_row0, Tq, pos0, slot, prefix_slot, prefix_len): query i of a sequence sits at position pos0+i and
    attends keys [0, pos0+i] (causal) or [0, pos0+Tq) (non-causal) of its slot / prefix slot.
    own_row_offset: the sequences' own keys start at row `own_row_offset` of their slots instead of row 0 (two-level prefixes: an image
    prefix keeps the rows in front for the system prompt it continues) - the own pools are handed over that many rows further on."""
    dt = _dt(q, k_cache, v_cache)
    out = torch.empty_like(q) if out is None else out
    k_prefix = k_cache if k_prefix is None else k_prefix
    v_prefix = v_cache if v_prefix is None else v_prefix
    off = int(own_row_offset) * D * k_cache.element_size()
    _lib.check(_lib_ready().vdd_flash_attention(q.data_ptr(), k_cache.data_ptr() + off, v_cache.data_ptr() + off, k_prefix.data_ptr(), v_prefix.data_ptr(),
                                                seqs.data_ptr(), out.data_ptr(), n_seq, max_tq, H, Hkv, D, k_cache.stride(0),
                                                k_cache.shape[2], k_prefix.stride(0), k_prefix.shape[2], D ** -0.5 if scale is None else float(scale),
                                                1 if causal else 0, dt, _st(q)))
    return out


def attention_probs(q, k_cache, seq, H, Hkv, D, k_prefix=None, scale=None):
    """The materialised attention map of ONE sequence: q [.., H*D] rotated queries, seq = (q_row0, Tq, pos0, slot, prefix_slot, prefix_len)
    host integers, keys from [prefix slot of k_prefix | own slot of k_cache].  Returns [H, Tq, pos0 + Tq] in q's dtype: fp32 softmax over the
    causal keys, rounded; zeros behind the diagonal (what HF's eager attention returns as attention weights)."""
    dt = _dt(q, k_cache)
    k_prefix = k_cache if k_prefix is None else k_prefix
    q_row0, Tq, pos0 = int(seq[0]), int(seq[1]), int(seq[2])
    out = torch.empty(H, Tq, pos0 + Tq, dtype=q.dtype, device=q.device)
    desc = (C.c_int32 * 6)(*[int(v) for v in seq])
    _lib.check(_lib_ready().vdd_attention_probs(q.data_ptr(), k_cache.data_ptr(), k_prefix.data_ptr(), C.cast(desc, C.c_void_p), out.data_ptr(), H, Hkv, D,
                                                k_cache.stride(0), k_cache.shape[2], k_prefix.stride(0), k_prefix.shape[2],
                                                D ** -0.5 if scale is None else float(scale), dt, _st(q)))
    return out


def flash_packs(seq_rows, max_per_pack=4):
    """Host side of the packed suffix pass: seq_rows = [(q_row0, Tq, pos0, slot, prefix_slot, prefix_len), ...] -> [[s0, s1, s2, s3], ...]
    (sequence indices, -1 = none): consecutive sequences continuing the same (prefix_slot, prefix_len), at most four per pack."""
    packs, cur, key = [], [], None
    for i, r in enumerate(seq_rows):
        k = (int(r[4]), int(r[5]))
        if cur and (k != key or len(cur) == max_per_pack):
            packs.append(cur + [-1] * (4 - len(cur))); cur = []
        cur.append(i); key = k
    if cur:
        packs.append(cur + [-1] * (4 - len(cur)))
    return packs


def flash_attention_packed(q, k_cache, v_cache, seqs, packs, n_packs, H, Hkv, D, out=None, k_prefix=None, v_prefix=None, scale=None):
    """flash_attention (causal) for sequences of at most 32 query rows that continue shared prefixes, four to a workgroup:
    packs int32 [n_packs, 4] from flash_packs().  The tiles inside the prefix are staged once per pack, not once per sequence."""
    dt = _dt(q, k_cache, v_cache)
    out = torch.empty_like(q) if out is None else out
    k_prefix = k_cache if k_prefix is None else k_prefix
    v_prefix = v_cache if v_prefix is None else v_prefix
    _lib.check(_lib_ready().vdd_flash_attention_packed(q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), k_prefix.data_ptr(), v_prefix.data_ptr(),
                                                       seqs.data_ptr(), packs.data_ptr(), out.data_ptr(), n_packs, H, Hkv, D, k_cache.stride(0),
                                                       k_cache.shape[2], k_prefix.stride(0), k_prefix.shape[2],
                                                       D ** -0.5 if scale is None else float(scale), dt, _st(q)))
    return out


def layernorm(x, w, b, eps, out=None):
    dt = _dt(x, w, b)
    M, d = x.shape
    out = torch.empty_like(x) if out is None else out
    _lib.check(_lib_ready().vdd_layernorm(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, d, eps, dt, _st(x)))
    return out


ACT_NONE, ACT_QUICK_GELU, ACT_GELU = 0, 1, 2


def bias_act(x, bias, act=ACT_NONE, out=None):
    """out = act(x + bias) rowwise; bias may be None."""
    dt = _dt(x, bias)
    M, d = x.shape
    out = torch.empty_like(x) if out is None else out
    _lib.check(_lib_ready().vdd_bias_act(x.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), M, d, act, dt, _st(x)))
    return out


_IMG_DT = {torch.float32: _lib.VDD_F32, torch.float16: _lib.VDD_F16, torch.bfloat16: _lib.VDD_BF16}


def vit_im2col(images, patch, k_pad, out=None, dtype=torch.bfloat16):
    """images [n, 3, S, S] (fp32 / fp16 / bf16, device) -> patches [n * (S/patch)^2, k_pad] of the model `dtype` (zero padded columns)."""
    if not images.is_cuda or images.dtype not in _IMG_DT or not images.is_contiguous():
        raise ValueError("vit_im2col takes a contiguous fp32 / fp16 / bf16 device tensor [n, 3, S, S]")
    n, _, S, _ = images.shape
    G = S // patch
    out = torch.empty(n * G * G, k_pad, dtype=dtype, device=images.device) if out is None else out
    dt = _dt(out)
    _lib.check(_lib_ready().vdd_vit_im2col(images.data_ptr(), _IMG_DT[images.dtype], out.data_ptr(), n, S, patch, k_pad, dt, _st(images)))
    return out


def vit_assemble(emb, cls, pos, n, T, out=None):
    """h[i, t] = (cls if t == 0 else emb[i * (T - 1) + t - 1]) + pos[t]  ->  [n * T, width]."""
    dt = _dt(emb, cls, pos)
    w = emb.shape[1]
    out = torch.empty(n * T, w, dtype=emb.dtype, device=emb.device) if out is None else out
    _lib.check(_lib_ready().vdd_vit_assemble(emb.data_ptr(), cls.data_ptr(), pos.data_ptr(), out.data_ptr(), n, T, w, dt, _st(emb)))
    return out


def vit_qkv_split(qkv, k_cache, v_cache, n, T, H, D, q_out=None, kv_only=False):
    """qkv [n * T, 3 * H * D] -> q [n * T, H * D]; K / V written to caches [>= n, H, t_max, D] (or views of them starting at a later
    token).  kv_only: the input is a fused [k, v] projection [n * T, 2 * H * D] (cross-attention); returns None."""
    dt = _dt(qkv, k_cache, v_cache)
    if not kv_only:
        q_out = torch.empty(n * T, H * D, dtype=qkv.dtype, device=qkv.device) if q_out is None else q_out
    _lib.check(_lib_ready().vdd_vit_qkv_split(qkv.data_ptr(), q_out.data_ptr() if not kv_only else None, k_cache.data_ptr(), v_cache.data_ptr(),
                                              n, T, H, D, k_cache.stride(0), k_cache.stride(1) // D, 2 if kv_only else 3, dt, _st(qkv)))
    return None if kv_only else q_out


def add(a, b, out=None):
    """out = a + b (one dtype, same shape, contiguous)."""
    dt = _dt(a, b)
    if a.shape != b.shape or not a.is_contiguous() or not b.is_contiguous():
        raise ValueError("add takes two contiguous tensors of one shape")
    out = torch.empty_like(a) if out is None else out
    _lib.check(_lib_ready().vdd_add(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), dt, _st(a)))
    return out


# ---- logits processors that need the token history (csrc/vdd_logits_process.hip) ------------------------------------------------
class StopWords:
    """Device-side form of a `stop_words_ids` list (qwen_generation_utils.py:318-345: sequences equal to [eos] are dropped)."""

    def __init__(self, stop_words_ids, eos_token_id: int, device):
        if not isinstance(stop_words_ids, list) or len(stop_words_ids) == 0:
            raise ValueError(f"`stop_words_ids` has to be a non-emtpy list, but is {stop_words_ids}.")
        if any(not isinstance(w, list) for w in stop_words_ids):
            raise ValueError(f"`stop_words_ids` has to be a list of lists, but is {stop_words_ids}.")
        import numpy as np
        if any(any((not isinstance(t, (int, np.integer)) or isinstance(t, bool) or t < 0) for t in w) for w in stop_words_ids):
            raise ValueError(f"Each list in `stop_words_ids` has to be a list of positive integers, but is {stop_words_ids}.")
        self.seqs = [[int(t) for t in w] for w in stop_words_ids if [int(t) for t in w] != [int(eos_token_id)]]
        assert all(len(w) > 0 for w in self.seqs), f"Stop words token sequences {stop_words_ids} cannot have an empty list"
        self.eos_token_id = int(eos_token_id)
        self.max_len = max([len(w) for w in self.seqs] + [1])
        off = [0]
        for w in self.seqs:
            off.append(off[-1] + len(w))
        self.flat = torch.tensor([t for w in self.seqs for t in w] or [0], dtype=torch.long, device=device)
        self.off = torch.tensor(off, dtype=torch.int32, device=device)

    def prompt_tail(self, prompts, device):
        """[Q, max_len] int64: the last max_len ids of every prompt, left-padded with -1."""
        L = self.max_len
        rows = [([-1] * L + [int(t) for t in r])[-L:] for r in prompts]
        return torch.tensor(rows, dtype=torch.long, device=device).reshape(len(prompts), L)


def stop_words_match(sw: StopWords, prompt_tail, gen, step=0, step_ptr=None, out=None):
    """out[q] = 1 iff (prompt q ++ its first `step (+ *step_ptr)` columns of gen) ends with one of sw's sequences."""
    Q = prompt_tail.shape[0]
    out = torch.empty(Q, dtype=torch.int32, device=prompt_tail.device) if out is None else out
    _lib.check(_lib_ready().vdd_stop_words_match(gen.data_ptr() if gen is not None else None, gen.stride(0) if gen is not None else 0,
                                                 int(step), step_ptr.data_ptr() if step_ptr is not None else None,
                                                 prompt_tail.data_ptr(), prompt_tail.shape[1], sw.flat.data_ptr(), sw.off.data_ptr(),
                                                 len(sw.seqs), out.data_ptr(), Q, _st(prompt_tail)))
    return out


_SCORE_DT = {torch.float32: _lib.VDD_F32, torch.float16: _lib.VDD_F16, torch.bfloat16: _lib.VDD_BF16}


def repetition_penalty_(scores, penalty, prompt_ids, gen, step=0, step_ptr=None, reciprocal=False):
    """HF RepetitionPenaltyLogitsProcessor in place on scores [Q, V]: history = prompt_ids [Q, Lp] (int64, -1 = padding) ++ the
    first `step (+ *step_ptr)` columns of gen."""
    if not scores.is_cuda or scores.dtype not in _SCORE_DT or scores.stride(1) != 1:
        raise ValueError("repetition_penalty_ takes a device [Q, V] tensor with contiguous rows")
    Q, V = scores.shape
    Lp = prompt_ids.shape[1] if prompt_ids is not None else 0
    _lib.check(_lib_ready().vdd_repetition_penalty(scores.data_ptr(), scores.stride(0), _SCORE_DT[scores.dtype], Q, V,
                                                   prompt_ids.data_ptr() if Lp else None, Lp,
                                                   gen.data_ptr() if gen is not None else None, gen.stride(0) if gen is not None else 0,
                                                   int(step), step_ptr.data_ptr() if step_ptr is not None else None, float(penalty),
                                                   _lib.TEMP_RECIPROCAL if reciprocal else 0, _st(scores)))
    return scores
