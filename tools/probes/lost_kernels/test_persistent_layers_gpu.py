"""Persistent few-row decode layers (vdd_layer_persistent.hip, lost_ops.decode_layers; laboratory code since round 6): ALL decoder layers of a 1 - 4 row decode step
in one launch, against the five-launch layer of the same package (vdd_skinny_gemm_normed + vdd_decode_attention_fused +
vdd_skinny_gemm_resid_ss + vdd_skinny_swiglu_normed + vdd_skinny_gemm_resid_ss) and an fp32 torch reference of the layer
(reference: one HF-eager LlamaDecoderLayer per branch and token, llava_llama.py:88-103; B = 1 per call, llava_calibrate.py:130)."""
import math

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.probe]
DEV = "cuda:0"


@pytest.fixture(autouse=True, params=[torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def storage_dtype(request):
    global DT
    DT = request.param
    yield


@pytest.fixture(scope="module")
def ops():
    import lost_ops as o          # decode_layers* + (through its __getattr__) the product's ops
    return o


def _rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, device=DEV, generator=g) * scale).to(DT)


def _rope_table(max_pos, D, theta=10000.0):
    inv = 1.0 / (theta ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    f = torch.outer(torch.arange(max_pos, dtype=torch.float32), inv)
    return torch.stack([f.cos(), f.sin()], -1).contiguous().to(DEV)          # [pos, D/2, 2]


def _setup(ops, M, d, H, F, n_layers, lens, plens, bias=False, seed=0):
    """Weights, KV pools with a random history, the step's rows.  Row r: prefix slot r % n_pre with plens[r] keys + own keys up to lens[r] - 1."""
    D = 128
    n_pre, t_pre, t_own = 2, max(plens) + 8, max(l - p for l, p in zip(lens, plens)) + 8
    L = []
    for i in range(n_layers):
        s = seed + 100 * i
        L.append(dict(ln1=(1 + _rnd(d, scale=0.1, seed=s + 1)), wqkv=_rnd(3 * d, d, scale=0.02, seed=s + 2),
                      bqkv=_rnd(3 * d, scale=0.1, seed=s + 3) if bias else None, wo=_rnd(d, d, scale=0.02, seed=s + 4),
                      ln2=(1 + _rnd(d, scale=0.1, seed=s + 5)), wgu=_rnd(2 * F, d, scale=0.02, seed=s + 6), wd=_rnd(d, F, scale=0.02, seed=s + 7),
                      k_own=_rnd(M, H, t_own, D, seed=s + 8), v_own=_rnd(M, H, t_own, D, seed=s + 9),
                      k_pre=_rnd(n_pre, H, t_pre, D, seed=s + 10), v_pre=_rnd(n_pre, H, t_pre, D, seed=s + 11)))
    i32 = dict(dtype=torch.int32, device=DEV)
    rows = torch.tensor([[r, lens[r], r % n_pre, plens[r]] for r in range(M)], **i32)
    pos = torch.tensor([lens[r] - 1 for r in range(M)], **i32)
    cpos = torch.tensor([lens[r] - 1 - plens[r] for r in range(M)], **i32)
    slot = torch.arange(M, **i32)
    x = _rnd(M, d, scale=0.5, seed=seed + 77)
    return L, rows, pos, cpos, slot, x, _rope_table(max(lens) + 4, D)


def _five_launch(ops, L, x, pos, cpos, slot, cs, rows, H, F, eps):
    """The package's own few-row layer chain (engine.LanguageModel._decode_step_few_rows before this round)."""
    D = 128
    resid, ss = x, None
    for l in L:
        if ss is None:
            qkv = ops.linear(ops.rmsnorm(resid, l["ln1"], eps), l["wqkv"], bias=l["bqkv"])
        else:
            qkv = ops.linear_normed(resid, ss, l["ln1"], eps, l["wqkv"], bias=l["bqkv"])
        att = ops.decode_attention_fused(qkv, pos, cpos, slot, cs, l["k_own"], l["v_own"], rows, H, H, D, k_prefix=l["k_pre"], v_prefix=l["v_pre"])
        resid, ss = ops.linear_resid_ss(att, l["wo"], resid)
        act = ops.swiglu_linear_normed(resid, ss, l["ln2"], eps, l["wgu"])
        resid, ss = ops.linear_resid_ss(act, l["wd"], resid)
    return resid, ss


def _persistent(ops, L, x, pos, cpos, slot, cs, rows, H, F, eps, ws=None):
    M, d = x.shape
    desc = ops.layer_descriptors(L, DEV)
    ws = ops.decode_layers_workspace(M, d, H, F, 128, DEV, DT) if ws is None else ws
    l0 = L[0]
    resid, ss = ops.decode_layers(desc, len(L), x, pos, cpos, slot, cs, rows, H, F, 128, eps, l0["k_own"].stride(0), l0["k_own"].shape[2],
                                  l0["k_pre"].stride(0), l0["k_pre"].shape[2], l0["bqkv"] is not None, ws)
    torch.cuda.synchronize()
    assert ops.decode_layers_status(ws) == 0, hex(ops.decode_layers_status(ws))
    return resid, ss, ws


def _clone_pools(L):
    return [{k: (v.clone() if k in ("k_own", "v_own") else v) for k, v in l.items()} for l in L]


@pytest.mark.parametrize("M,n_layers,bias", [(2, 1, False), (1, 1, False), (2, 3, False), (3, 2, False), (4, 2, True), (2, 2, True)])
def test_persistent_layers_equal_the_five_launch_layer_at_7b_widths(ops, M, n_layers, bias):
    d, H, F, eps = 4096, 32, 11008, 1e-5
    if ops.decode_layers_max_rows(d, H, H, F, 128, n_layers, DT) < M:
        pytest.skip("shape not served on this device")
    lens = [650, 75, 333, 18][:M]
    plens = [611, 36, 300, 0][:M]
    L, rows, pos, cpos, slot, x, cs = _setup(ops, M, d, H, F, n_layers, lens, plens, bias=bias, seed=3)
    La, Lb = _clone_pools(L), _clone_pools(L)
    ra, sa = _five_launch(ops, La, x, pos, cpos, slot, cs, rows, H, F, eps)
    rb, sb, _ = _persistent(ops, Lb, x, pos, cpos, slot, cs, rows, H, F, eps)
    # same rounding points, different accumulation order: a few ulps of the 16-bit type on O(1) values
    ulp = 2.0 ** -7 if DT == torch.bfloat16 else 2.0 ** -10
    err = (ra.float() - rb.float()).abs().max().item()
    assert err <= 6 * ulp * max(1.0, ra.float().abs().max().item()), err
    ssa, ssb = sa.sum(1), sb.sum(1)
    assert torch.allclose(ssa, ssb, rtol=2e-2), (ssa, ssb)
    # the new token's K / V landed in the cache of every layer, identically (same RoPE arithmetic)
    for la, lb in zip(La, Lb):
        for r in range(M):
            cp = int(cpos[r])
            assert (la["k_own"][r, :, cp].float() - lb["k_own"][r, :, cp].float()).abs().max().item() <= 8 * ulp * 4
            assert (la["v_own"][r, :, cp].float() - lb["v_own"][r, :, cp].float()).abs().max().item() <= 8 * ulp * 4
        untouched = torch.ones_like(la["k_own"], dtype=torch.bool)
        for r in range(M):
            untouched[r, :, int(cpos[r])] = False
        assert torch.equal(lb["k_own"][untouched], L[0]["k_own"][untouched]) if la is La[0] else True


def test_persistent_layers_replay_with_fresh_epochs_and_are_deterministic(ops):
    """Ten launches on one workspace (what a captured graph replays): identical results every time - stale granules of an earlier
    launch never satisfy a later one (epochs come from the launch counter in the workspace)."""
    d, H, F, eps, M, n_layers = 4096, 32, 11008, 1e-5, 2, 2
    if ops.decode_layers_max_rows(d, H, H, F, 128, n_layers, DT) < M:
        pytest.skip("shape not served on this device")
    L, rows, pos, cpos, slot, x, cs = _setup(ops, M, d, H, F, n_layers, [400, 60], [350, 30], seed=9)
    r0, s0, ws = _persistent(ops, _clone_pools(L), x, pos, cpos, slot, cs, rows, H, F, eps)
    for i in range(9):
        xi = x if i % 2 == 0 else (x.float() * 0.5).to(DT)          # alternate inputs: a stale hand-off would show
        ri, si, _ = _persistent(ops, _clone_pools(L), xi, pos, cpos, slot, cs, rows, H, F, eps, ws=ws)
        if i % 2 == 0:
            assert torch.equal(ri, r0) and torch.equal(si, s0), i
        else:
            assert not torch.equal(ri, r0)
    assert int(ws[:4].view(torch.int32).item()) == 10


def test_persistent_layers_against_fp32_torch_layer(ops):
    """One layer at 7B widths against the layer written out in fp32 torch on the rounded weights (the reference's arithmetic:
    RMSNorm, rotate-half RoPE, softmax attention over [prefix | own | new] keys, SwiGLU)."""
    d, H, F, eps, M, D = 4096, 32, 11008, 1e-5, 2, 128
    if ops.decode_layers_max_rows(d, H, H, F, D, 1, DT) < M:
        pytest.skip("shape not served on this device")
    lens, plens = [200, 40], [150, 20]
    L, rows, pos, cpos, slot, x, cs = _setup(ops, M, d, H, F, 1, lens, plens, seed=21)
    Lb = _clone_pools(L)
    rb, sb, _ = _persistent(ops, Lb, x, pos, cpos, slot, cs, rows, H, F, eps)
    l = {k: (v.float() if v is not None else None) for k, v in L[0].items()}
    out = []
    for r in range(M):
        h = x[r].float()
        a = h * torch.rsqrt((h * h).mean() + eps) * l["ln1"]
        qkv = l["wqkv"] @ a
        q, k, v = qkv[:d].view(H, D), qkv[d:2 * d].view(H, D), qkv[2 * d:].view(H, D)
        c, s = cs[int(pos[r]), :, 0], cs[int(pos[r]), :, 1]
        rot = lambda t: torch.cat([t[:, :64] * c - t[:, 64:] * s, t[:, 64:] * c + t[:, :64] * s], 1)
        q, k = rot(q), rot(k)
        n_own = lens[r] - 1 - plens[r]
        K = torch.cat([l["k_pre"][r % 2, :, :plens[r]], l["k_own"][r, :, :n_own], k[:, None]], 1)
        V = torch.cat([l["v_pre"][r % 2, :, :plens[r]], l["v_own"][r, :, :n_own], v[:, None]], 1)
        p = torch.softmax(torch.einsum("hd,htd->ht", q, K) / math.sqrt(D), -1)
        att = torch.einsum("ht,htd->hd", p, V).reshape(d)
        h = h + l["wo"] @ att
        a = h * torch.rsqrt((h * h).mean() + eps) * l["ln2"]
        gu = l["wgu"] @ a
        h = h + l["wd"] @ (torch.nn.functional.silu(gu[:F]) * gu[F:])
        out.append(h)
    want = torch.stack(out)
    tol = 0.06 if DT == torch.bfloat16 else 0.01
    assert (rb.float() - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())
    assert torch.allclose(sb.sum(1), (want * want).sum(1), rtol=3e-2)
