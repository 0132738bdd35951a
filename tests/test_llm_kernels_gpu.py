"""Numerics of the hand-written model kernels vs plain PyTorch fp32 references of the same op
(bf16 inputs, so tolerances are bf16-rounding sized and written next to each check)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DT = torch.bfloat16          # storage type of the current test run (set by the fixture below)


@pytest.fixture(autouse=True, params=[torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def storage_dtype(request):
    """Every test of this file runs once per storage type of the model kernels (csrc/vdd_elem.h): bf16 (BASELINE config #2) and fp16
    (what the reference's drivers load, builder.py:40).  The tolerances are written for bf16 (8 significant bits); fp16 (11) sits inside them."""
    global DT
    DT = request.param
    yield
    DT = torch.bfloat16




def ops():
    from llava_align_amd import ops as O
    return O


def bf(*shape, scale=1.0, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, device=DEV, generator=g) * scale).to(DT)


@pytest.mark.parametrize("M,d", [(1, 4096), (7, 4096), (3, 5120), (5, 1024), (2, 256)])
def test_rmsnorm(M, d):
    O = ops()
    x, dl, w = bf(M, d, seed=1), bf(M, d, seed=2), bf(d, seed=3) * 0.1 + 1
    ro = torch.empty_like(x)
    y = O.rmsnorm(x, w, 1e-5, delta=dl, resid_out=ro)
    h = (x.float() + dl.float()).to(DT)
    assert torch.equal(ro, h)
    ref = (h.float() * torch.rsqrt(h.float().pow(2).mean(-1, keepdim=True) + 1e-5)).to(DT).float() * w.float()
    assert torch.allclose(y.float(), ref, rtol=1.6e-2, atol=1e-3)       # 2 bf16 roundings
    y2 = O.rmsnorm(x, w, 1e-5)
    ref2 = (x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-5)).to(DT).float() * w.float()
    assert torch.allclose(y2.float(), ref2, rtol=1.6e-2, atol=1e-3)


def rope_table(max_pos, D, theta=10000.0):
    inv = 1.0 / (theta ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    ang = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.stack([ang.cos(), ang.sin()], dim=-1).contiguous().to(DEV)      # [max_pos, D/2, 2]


def test_rope_kv_write():
    O = ops()
    M, H, Hkv, D, T, S = 6, 8, 4, 128, 32, 5
    qkv = bf(M, (H + 2 * Hkv) * D, seed=4)
    pos = torch.tensor([0, 3, 31, 7, 7, 12], dtype=torch.int32, device=DEV)
    slot = torch.tensor([0, 1, 2, 3, 4, 0], dtype=torch.int32, device=DEV)
    kc = torch.zeros(S, Hkv, T, D, dtype=DT, device=DEV)
    vc = torch.zeros_like(kc)
    cs = rope_table(64, D)
    q = O.rope_kv_write(qkv, pos, slot, cs, kc, vc, H, Hkv, D).view(M, H, D)

    def rot(x, p):      # x [.., D]
        c, s = cs[p.long(), :, 0], cs[p.long(), :, 1]
        a, b = x[..., : D // 2].float(), x[..., D // 2:].float()
        return torch.cat([a * c[:, None] - b * s[:, None], b * c[:, None] + a * s[:, None]], -1)
    qr = rot(qkv[:, : H * D].view(M, H, D), pos)
    kr = rot(qkv[:, H * D: (H + Hkv) * D].view(M, Hkv, D), pos)
    assert torch.allclose(q.float(), qr, rtol=8e-3, atol=8e-3)
    for m in range(M):
        assert torch.allclose(kc[slot[m], :, pos[m]].float(), kr[m], rtol=8e-3, atol=8e-3)
        assert torch.equal(vc[slot[m], :, pos[m]], qkv[m, (H + Hkv) * D:].view(Hkv, D))
    assert int((kc != 0).any(-1).sum()) == M * Hkv      # nothing else touched


def test_embed_scatter_writes_rows_in_place():
    O = ops()
    table = bf(1000, 256, seed=61)
    out = torch.full((40, 256), 7.0, dtype=DT, device=DEV)
    ids = torch.tensor([5, 999, 0, 5], dtype=torch.int32, device=DEV)
    rows = torch.tensor([3, 0, 39, 17], dtype=torch.int32, device=DEV)
    O.embed_scatter(ids, rows, table, out)
    want = torch.full((40, 256), 7.0, dtype=DT, device=DEV)
    want[rows.long()] = table[ids.long()]
    assert torch.equal(out, want)


def test_silu_mul_and_embed():
    O = ops()
    gu = bf(5, 2 * 11008, seed=5)
    y = O.silu_mul(gu)
    g, u = gu[:, :11008].float(), gu[:, 11008:].float()
    ref = torch.nn.functional.silu(g).to(DT).float() * u
    assert torch.allclose(y.float(), ref, rtol=1.6e-2, atol=1e-3)
    table = bf(1000, 4096, seed=6)
    ids = torch.tensor([0, 999, 5, 5, 17], device=DEV)
    assert torch.equal(O.embed(ids, table), table[ids])


@pytest.mark.parametrize("M,N,K", [(1, 4096, 4096), (2, 12288, 4096), (3, 4096, 11008), (16, 4096, 4096), (17, 32000, 4096),
                                    (33, 22016, 4096), (64, 4096, 11008), (5, 1000, 256), (4, 40, 128),
                                    (2, 15360, 5120), (3, 5120, 13824), (8, 27648, 5120), (6, 5120, 5120),      # this row: LLaVA-1.5-13B dims
                                    (24, 4096, 11008), (32, 4096, 4096), (18, 5120, 13824), (48, 4096, 4096), (40, 5120, 5120)])    # 17 - 64 rows, d-wide: eight-wave blocks
def test_skinny_gemm(M, N, K):
    O = ops()
    x, w = bf(M, K, seed=7), bf(N, K, scale=0.02, seed=8)
    r = bf(M, N, seed=9)
    y = O.skinny_gemm(x, w)
    ref = x.float() @ w.float().t()
    tol = 2e-2 * ref.abs().max().item()
    assert (y.float() - ref).abs().max().item() <= tol          # fp32 accumulate, one bf16 rounding (+ ordering)
    y2 = O.skinny_gemm(x, w, resid=r)
    ref2 = ref.to(DT).float() + r.float()
    assert (y2.float() - ref2).abs().max().item() <= 2e-2 * ref2.abs().max().item()
    for ns in (2, 4):                                       # split-K slabs (consumed by rmsnorm)
        if K % (128 * ns) == 0:
            sl = O.skinny_gemm(x, w, n_split=ns, slabs=True)
            assert sl.shape == (ns, M, N) and (sl.sum(0) - ref).abs().max().item() <= 1e-3 * ref.abs().max().item() + 1e-3
    # strided input view (rows of a bigger buffer)
    big = bf(M, K + 128, seed=10)
    y3 = O.skinny_gemm(big[:, :K], w)
    assert (y3.float() - big[:, :K].float() @ w.float().t()).abs().max().item() <= tol


@pytest.mark.parametrize("M", [17, 31, 32, 33, 48, 50, 64])
def test_wide_weight_streaming_17_to_64_rows(M):
    """skinny_wide_kernel: 32-column blocks, eight waves split K (ragged tails: 11008 / 8 = 1376 = 10 batches + 96; 13824 / 8), residual
    add, the SwiGLU form (16 gate + 16 up columns per block), N not a multiple of 32."""
    O = ops()
    for N, K in ((12288, 4096), (4096, 11008), (5120, 13824), (1000, 256), (40, 512)):
        x, w, r = bf(M, K, seed=81), bf(N, K, scale=0.02, seed=82), bf(M, N, seed=83)
        ref = x.float() @ w.float().t()
        y = O.skinny_gemm(x, w)
        assert (y.float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
        y2 = O.skinny_gemm(x, w, resid=r)
        ref2 = ref.to(DT).float() + r.float()
        assert (y2.float() - ref2).abs().max().item() <= 2e-2 * ref2.abs().max().item()
        if M <= O.skinny_rows(N, K):
            assert torch.equal(O.linear(x, w), y)                           # ops.linear routes these rows here
    for F, K in ((11008, 4096), (13824, 5120), (520, 256)):
        x, w = bf(M, K, seed=84), bf(2 * F, K, seed=85) * 0.05
        got = O.swiglu_linear(x, w)
        g, u = (x.float() @ w[:F].float().t()).to(DT).float(), (x.float() @ w[F:].float().t()).to(DT).float()
        ref = torch.nn.functional.silu(g).to(DT).float() * u
        assert got.shape == (M, F) and torch.allclose(got.float(), ref, rtol=3e-2, atol=3e-2)
        two = O.silu_mul(O.gemm(x, w)) if K % 128 == 0 else None           # the MFMA GEMM + SiLU*mul: same roundings, another k order
        if two is not None:
            assert (got.float() - two.float()).abs().max().item() <= 2 ** -6 * ref.abs().max().item()


def test_skinny_gemm_eight_wave_blocks_for_narrow_outputs():
    """N <= 8192 without slabs: eight waves per block split K eight ways (o / down projections of one question in flight);
    same result up to the summation order as the four-wave kernel on a wider N made of the same rows."""
    O = ops()
    for M, N, K in ((2, 4096, 4096), (3, 4096, 11008), (2, 5120, 13824), (8, 1024, 256)):
        x, w = bf(M, K, seed=31), bf(N, K, scale=0.02, seed=32)
        y = O.skinny_gemm(x, w)
        ref = x.float() @ w.float().t()
        assert (y.float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
        wide = torch.cat([w, w, w])[:8208]                              # N > 8192: the four-wave kernel
        y4 = O.skinny_gemm(x, wide)[:, :N]
        assert (y.float() - y4.float()).abs().max().item() <= 1e-2 * ref.abs().max().item()


def attn_ref(q, K, V):      # q [H, D], K/V [H, T, D] fp32
    s = torch.einsum("hd,htd->ht", q, K) / math.sqrt(q.shape[-1])
    return torch.einsum("ht,htd->hd", s.softmax(-1), V)


def test_decode_attention_ragged_and_prefix_shared():
    O = ops()
    H, Hkv, D, T, S = 8, 4, 128, 700, 6
    kc, vc = bf(S, Hkv, T, D, seed=11), bf(S, Hkv, T, D, seed=12)
    rows = torch.tensor([[0, 1, 0, 0], [1, 4, 0, 0], [2, 37, 0, 0], [3, 700, 0, 0],      # own slot only
                         [4, 650, 5, 611], [3, 620, 5, 611], [1, 5, 2, 4]], dtype=torch.int32, device=DEV)
    M = rows.shape[0]
    q = bf(M, H * D, seed=13)
    out = O.decode_attention(q, kc, vc, rows, H, Hkv, D).view(M, H, D)
    rep = H // Hkv
    for m, (slot, ln, ps, pl) in enumerate(rows.tolist()):
        K = torch.cat([kc[ps, :, :pl], kc[slot, :, :ln - pl]], 1).float().repeat_interleave(rep, 0)     # own slot is compact: token t at t - plen
        V = torch.cat([vc[ps, :, :pl], vc[slot, :, :ln - pl]], 1).float().repeat_interleave(rep, 0)
        ref = attn_ref(q[m].view(H, D).float(), K, V)
        assert torch.allclose(out[m].float(), ref, rtol=2e-2, atol=2e-2), m

@pytest.mark.parametrize("n_split", [1, 2, 4])
@pytest.mark.parametrize("H,Hkv", [(8, 8), (8, 2)])
def test_decode_attention_fused_equals_rope_write_then_attention(H, Hkv, n_split):
    """Small-M kernel (RoPE + KV write + whole-context attention + merge in one launch) vs the three-kernel path it
    replaces: identical cache contents (bit-exact: same RoPE rounding), outputs equal up to the softmax partition order.
    n_split > 1: the keys of a (row, head) cut over 2 / 4 workgroups, merged by the last one to finish (contexts of 0, 1, 65, 699 and
    649 old keys: empty slices, a slice holding only the new token, ragged last rounds); launched repeatedly - the tickets must return
    to zero - and bit-identical from launch to launch (the merge order is the slice order, not the arrival order)."""
    O = ops()
    D, T, S, TP = 128, 720, 5, 640
    cs = rope_table(800, D)
    kc, vc = bf(S, Hkv, T, D, seed=31), bf(S, Hkv, T, D, seed=32)
    kp, vp = bf(2, Hkv, TP, D, seed=33), bf(2, Hkv, TP, D, seed=34)
    # (slot, len, pslot, plen): len counts the new token; own index = len - 1 - plen
    rows = torch.tensor([[0, 1, 0, 0], [1, 2, 0, 0], [2, 66, 0, 0], [3, 700, 0, 0], [4, 650, 1, 611]], dtype=torch.int32, device=DEV)
    M = rows.shape[0]
    qkv = bf(M, (H + 2 * Hkv) * D, seed=35)
    pos = (rows[:, 1] - 1).contiguous()
    cpos = (rows[:, 1] - 1 - rows[:, 3]).contiguous()
    slot = rows[:, 0].contiguous()
    k1, v1, k2, v2 = kc.clone(), vc.clone(), kc.clone(), vc.clone()
    q = O.rope_kv_write(qkv, pos, slot, cs, k1, v1, H, Hkv, D, cpos=cpos)
    want = O.decode_attention(q, k1, v1, rows, H, Hkv, D, k_prefix=kp, v_prefix=vp)
    got = O.decode_attention_fused(qkv, pos, cpos, slot, cs, k2, v2, rows, H, Hkv, D, k_prefix=kp, v_prefix=vp, n_split=n_split)
    assert torch.equal(k1, k2) and torch.equal(v1, v2)
    assert torch.allclose(got.float(), want.float(), rtol=2e-2, atol=2e-2), (got.float() - want.float()).abs().max().item()
    for _ in range(20):
        again = O.decode_attention_fused(qkv, pos, cpos, slot, cs, k2, v2, rows, H, Hkv, D, k_prefix=kp, v_prefix=vp, n_split=n_split)
        assert torch.equal(again, got)

@pytest.mark.parametrize("M", [1, 2, 7, 8])
def test_swiglu_linear_equals_gemm_then_silu_mul(M):
    """Fused gate/up GEMV + SiLU*mul == vdd_skinny_gemm followed by vdd_silu_mul, bit for bit (same k order, same bf16
    rounding points); a feature count that is not a multiple of the 8-feature block is checked against torch."""
    O = ops()
    for F, K in ((11008, 4096), (520, 256), (516, 256)):
        x = bf(M, K, seed=41)
        w = bf(2 * F, K, seed=42) * 0.05
        meas = O.swiglu_linear(x, w)                       # whichever form was measured faster for this shape (ops._pick_form) ...
        O.FORCE_FORM["swiglu"] = "skinny"                  # ... and the fused weight-streaming kernel itself, which the bit-level statements are about
        try:
            got = O.swiglu_linear(x, w)
        finally:
            O.FORCE_FORM.pop("swiglu")
        assert got.shape == (M, F) and meas.shape == (M, F)
        assert (meas.float() - got.float()).abs().max().item() <= 2 ** -6 * got.float().abs().max().item() + 1e-3
        if F % 8 == 0:
            two = O.silu_mul(O.skinny_gemm(x, w))
            if 2 * F > 8192:
                assert torch.equal(got, two)
            else:       # narrow outputs: skinny_gemm splits K over EIGHT waves, the fused kernel over four - another fp32 summation order,
                        # visible as an ulp of the output here and there (fp16 keeps 11 bits of it, bf16 8)
                assert (got.float() - two.float()).abs().max().item() <= 2 ** -9 * two.float().abs().max().item()
        ref = torch.nn.functional.silu((x.float() @ w[:F].float().t()).to(DT).float()).to(DT).float() \
            * (x.float() @ w[F:].float().t()).to(DT).float()
        assert torch.allclose(got.float(), ref, rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("D,H,Hkv,causal", [(128, 4, 4, True), (128, 8, 2, True), (64, 4, 4, False)])
def test_flash_attention_prefill(D, H, Hkv, causal):
    """Packed sequences, ragged lengths (several 128-row query blocks with causally skipped tiles, a single row), sequences that
    continue a shared prefix held in another slot (a K / V tile then straddles the two pools)."""
    O = ops()
    T, S = 700, 4
    kc, vc = bf(S, Hkv, T, D, seed=21), bf(S, Hkv, T, D, seed=22)
    # (q_row0, Tq, pos0, slot, pslot, plen)
    seqs = [(0, 300, 0, 0, 0, 0), (300, 1, 0, 1, 0, 0), (301, 65, 0, 2, 0, 0), (366, 30, 611, 3, 0, 611), (396, 130, 36, 1, 2, 36)] if causal else \
           [(0, 577, 0, 0, 0, 0), (577, 33, 0, 1, 0, 0)]
    Ttot = sum(s[1] for s in seqs)
    q = bf(Ttot, H * D, seed=23)
    sd = torch.tensor(seqs, dtype=torch.int32, device=DEV)
    out = O.flash_attention(q, kc, vc, sd, len(seqs), max(s[1] for s in seqs), H, Hkv, D, causal=causal).view(Ttot, H, D)
    rep = H // Hkv
    for (r0, Tq, p0, slot, ps, pl) in seqs:
        Tk = p0 + Tq
        K = torch.cat([kc[ps, :, :pl], kc[slot, :, :Tk - pl]], 1).float().repeat_interleave(rep, 0)     # [H, Tk, D]; own slot compact
        V = torch.cat([vc[ps, :, :pl], vc[slot, :, :Tk - pl]], 1).float().repeat_interleave(rep, 0)
        Q = q[r0:r0 + Tq].view(Tq, H, D).float().transpose(0, 1)                                     # [H, Tq, D]
        s = Q @ K.transpose(1, 2) / math.sqrt(D)
        if causal:
            mask = torch.arange(Tk, device=DEV)[None, :] > (p0 + torch.arange(Tq, device=DEV))[:, None]
            s = s.masked_fill(mask[None], -float("inf"))
        ref = (s.softmax(-1) @ V).transpose(0, 1)
        got = out[r0:r0 + Tq].float()
        assert torch.allclose(got, ref, rtol=2e-2, atol=2e-2), (r0, (got - ref).abs().max().item())


def test_flash_attention_packed_equals_one_sequence_per_block():
    """The suffix pass packs four short sequences of one prefix into a workgroup (shared prefix tiles staged once): same tiles in the
    same order per wave, so the result is bit-identical to the one-sequence-per-block launch."""
    O = ops()
    H, Hkv, D = 8, 4, 128
    kp, vp = bf(3, Hkv, 640, D, seed=61), bf(3, Hkv, 640, D, seed=62)                 # prefixes: slot 0 (611 keys), slot 1 (36), slot 2 (100)
    ko, vo = bf(16, Hkv, 64, D, seed=63), bf(16, Hkv, 64, D, seed=64)
    rows, r = [], 0
    for i, (tq, ps, pl) in enumerate([(25, 0, 611), (19, 0, 611), (28, 0, 611), (32, 0, 611), (1, 0, 611), (22, 0, 611),      # six of one image
                                      (24, 1, 36), (20, 1, 36), (27, 1, 36),                                                # a short prefix: no shared tile
                                      (30, 2, 100), (17, 2, 100),                                                            # one shared tile + a straddling one
                                      (9, 0, 0)]):                                                                           # no prefix at all
        rows.append([r, tq, pl, i, ps, pl]); r += tq
    q = bf(r, H * D, seed=65)
    sd = torch.tensor(rows, dtype=torch.int32, device=DEV)
    a = O.flash_attention(q, ko, vo, sd, len(rows), 32, H, Hkv, D, causal=True, k_prefix=kp, v_prefix=vp)
    packs = O.flash_packs(rows)
    assert packs == [[0, 1, 2, 3], [4, 5, -1, -1], [6, 7, 8, -1], [9, 10, -1, -1], [11, -1, -1, -1]]
    b = O.flash_attention_packed(q, ko, vo, sd, torch.tensor(packs, dtype=torch.int32, device=DEV), len(packs), H, Hkv, D, k_prefix=kp, v_prefix=vp)
    assert torch.equal(a, b)


def test_layernorm_and_bias_act():
    O = ops()
    x, w, b = bf(9, 1024, seed=31), bf(1024, seed=32) * 0.1 + 1, bf(1024, seed=33) * 0.1
    y = O.layernorm(x, w, b, 1e-5)
    ref = torch.nn.functional.layer_norm(x.float(), (1024,), w.float(), b.float(), 1e-5)
    assert torch.allclose(y.float(), ref, rtol=1e-2, atol=1e-2)
    z = bf(7, 4096, seed=34)
    bias = bf(4096, seed=35)
    zb = (z.float() + bias.float()).to(DT).float()
    assert torch.allclose(O.bias_act(z, bias, O.ACT_QUICK_GELU).float(), zb * torch.sigmoid(1.702 * zb), rtol=1e-2, atol=1e-2)
    assert torch.allclose(O.bias_act(z, bias, O.ACT_GELU).float(), torch.nn.functional.gelu(zb), rtol=1e-2, atol=1e-2)
    assert torch.equal(O.bias_act(z, bias, O.ACT_NONE), zb.to(DT))
    assert torch.equal(O.bias_act(z, None, O.ACT_NONE), z)


def test_grouped_prefix_decode_attention_equals_per_row():
    """Rows sharing a prompt prefix attended as a group (MFMA over the group's queries) == one query at a time."""
    O = ops()
    H, Hkv, D = 8, 4, 128
    kp, vp = bf(3, Hkv, 640, D, seed=41), bf(3, Hkv, 640, D, seed=42)          # prefix pool
    ko, vo = bf(40, Hkv, 128, D, seed=43), bf(40, Hkv, 128, D, seed=44)        # compact own slots
    rows, groups, grp_rows = [], [], []
    # group 0: 6 rows on prefix slot 0 (611 tokens); group 1: 21 rows on prefix slot 2 (36 tokens); 2 rows without prefix
    for i in range(6):
        rows.append([i, 611 + 20 + i, 0, 611])
    groups.append([0, 6, 0, 611]); grp_rows += list(range(0, 6))
    for i in range(21):
        rows.append([6 + i, 36 + 5 + 3 * i, 2, 36])
    groups.append([6, 21, 2, 36]); grp_rows += list(range(6, 27))
    rows += [[27, 90, 0, 0], [28, 1, 0, 0]]
    # shuffle the row order so that group membership is a real gather
    perm = torch.randperm(len(rows), generator=torch.Generator().manual_seed(0)).tolist()
    inv = {old: new for new, old in enumerate(perm)}
    rows_p = [rows[o] for o in perm]
    grp_rows_p = [inv[r] for r in grp_rows]
    M = len(rows)
    q = bf(M, H * D, seed=45)
    rt = torch.tensor(rows_p, dtype=torch.int32, device=DEV)
    a = O.decode_attention(q, ko, vo, rt, H, Hkv, D, k_prefix=kp, v_prefix=vp, max_len=768)
    items = O.prefix_work_items(groups)
    with pytest.raises(ValueError):                     # the prefix pass needs the fragment-major image of the prefix K / V
        O.decode_attention_grouped(q, ko, vo, kp, vp, rt, torch.tensor(groups, dtype=torch.int32, device=DEV),
                                   torch.tensor(grp_rows_p, dtype=torch.int32, device=DEV),
                                   torch.tensor(items, dtype=torch.int32, device=DEV), len(items), H, Hkv, D, 611, 128)
    # MFMA prefix pass on the fragment-major image: per 64-key chunk 16 K fragments then 16 V^T fragments of 1 KiB (lane-linear)
    pf = torch.full((kp.shape[0], Hkv, 2 * 640, D), float("nan"), dtype=DT, device=DEV)
    plen_of_slot = torch.tensor([611, 0, 36], dtype=torch.int32, device=DEV)
    O.prefix_fragments(kp, vp, pf, plen_of_slot)
    blk = pf[0].view(Hkv, 10, 2, 16, 64, 8)             # [head][chunk][K | V^T][fragment][lane][8]
    ln, g = torch.arange(64, device=DEV) % 16, torch.arange(64, device=DEV) // 16
    kz, vz = kp[0].clone(), vp[0].clone()
    kz[:, 611:], vz[:, 611:] = 0, 0                     # keys past the prefix are zero-filled in the image
    for t in range(4):
        key = (t >> 1) * 32 + (ln >> 2) * 8 + (t & 1) * 4 + (ln & 3)                 # MFMA row -> key of tile t
        for ks in range(4):
            want = torch.stack([kz[:, c * 64 + key].view(Hkv, 64, 4, 4, 8)[:, torch.arange(64), ks, g] for c in range(10)], 1)
            assert torch.equal(blk[:, :, 0, t * 4 + ks], want), (t, ks)
    for kk in range(2):
        for nt in range(8):
            rows = (kk * 32 + g * 8)[:, None] + torch.arange(8, device=DEV)[None]     # [lane][8 keys]
            want = torch.stack([vz[:, c * 64 + rows, (ln * 8 + nt)[:, None]] for c in range(10)], 1)          # column ln of tile nt = dim 8 ln + nt
            assert torch.equal(blk[:, :, 1, kk * 8 + nt], want), (kk, nt)
    assert torch.isnan(pf[1].float()).all() and not torch.isnan(pf[2, :, :128].float()).any()     # empty slot untouched; 36 keys = 1 chunk
    c = O.decode_attention_grouped(q, ko, vo, kp, vp, rt, torch.tensor(groups, dtype=torch.int32, device=DEV),
                                   torch.tensor(grp_rows_p, dtype=torch.int32, device=DEV),
                                   torch.tensor(items, dtype=torch.int32, device=DEV), len(items), H, Hkv, D, 611, 128, prefix_frag=pf)
    assert torch.allclose(a.float(), c.float(), rtol=3e-2, atol=3e-2)
    # several 64-key chunks per work item (online softmax inside the wave, one partial per item): 2 -> 5 items cover 611 keys,
    # 4 -> 3 items with a ragged last one, 16 -> the whole prefix in one item
    for cpi in (2, 4, 16):
        it = O.prefix_work_items(groups, cpi)
        assert len(it) == sum(-(-n // 16) * -(-pl // (64 * cpi)) for _, n, _, pl in groups)
        cc = O.decode_attention_grouped(q, ko, vo, kp, vp, rt, torch.tensor(groups, dtype=torch.int32, device=DEV),
                                        torch.tensor(grp_rows_p, dtype=torch.int32, device=DEV),
                                        torch.tensor(it, dtype=torch.int32, device=DEV), len(it), H, Hkv, D, 611, 128, prefix_frag=pf,
                                        chunks_per_item=cpi)
        assert torch.allclose(a.float(), cc.float(), rtol=3e-2, atol=3e-2), cpi
    # own ranges declared longer than 256 keys keep the split-KV own pass + combine; shorter ones finish in one wave per (row, head)
    it = O.prefix_work_items(groups, 4)
    long_own = O.decode_attention_grouped(q, ko, vo, kp, vp, rt, torch.tensor(groups, dtype=torch.int32, device=DEV),
                                          torch.tensor(grp_rows_p, dtype=torch.int32, device=DEV),
                                          torch.tensor(it, dtype=torch.int32, device=DEV), len(it), H, Hkv, D, 611, 320, prefix_frag=pf,
                                          chunks_per_item=4)
    assert torch.allclose(a.float(), long_own.float(), rtol=3e-2, atol=3e-2)
    b = cc
    rep = H // Hkv
    for m, (slot, ln, ps, pl) in enumerate(rows_p):
        K = torch.cat([kp[ps, :, :pl], ko[slot, :, :ln - pl]], 1).float().repeat_interleave(rep, 0)
        V = torch.cat([vp[ps, :, :pl], vo[slot, :, :ln - pl]], 1).float().repeat_interleave(rep, 0)
        ref = attn_ref(q[m].view(H, D).float(), K, V)
        assert torch.allclose(b[m].view(H, D).float(), ref, rtol=2e-2, atol=2e-2), m


def _gemm_ref(x, w, epi, bias, resid):
    """Plain fp32 PyTorch restatement of vdd_gemm's epilogues, bf16 rounding where the HF modules round."""
    O = ops()
    acc = x.float() @ w.float().t()
    r = lambda t: t.to(DT).float()
    if epi == O.EPI_NONE:
        return r(acc)
    if epi == O.EPI_SWIGLU:
        F = w.shape[0] // 2
        g, u = r(acc[:, :F]), r(acc[:, F:])
        return r(r(g / (1 + torch.exp(-g))) * u)
    y = r(acc + bias.float())
    if epi == O.EPI_BIAS:
        return y
    if epi == O.EPI_BIAS_QUICK_GELU:
        return r(y / (1 + torch.exp(-1.702 * y)))
    if epi == O.EPI_BIAS_GELU:
        return r(torch.nn.functional.gelu(y))
    return r(y + resid.float())


# (M, N, K): the decode batch, a ragged prefill chunk, ViT / projector shapes, LLaVA-1.5-13B (d 5120, ffn 13824), Qwen's vocabulary
GEMM_SHAPES = [(9, 4096, 4096), (77, 136, 128), (300, 520, 256), (768, 12288, 4096), (768, 4096, 11008), (1000, 2304, 1024), (577, 1024, 640),
               (96, 15360, 5120), (96, 5120, 13824), (40, 151936, 256), (2111, 4096, 4096)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_matches_fp32_reference(M, N, K):
    """Every macro-tile shape and schedule (hybrid, data-parallel only, stream-K only) of the persistent MFMA GEMM gives the same
    product; tolerance = 2 bf16 ulps of the largest output (bf16 output rounding + fp32 accumulation-order differences)."""
    O = ops()
    x, w = bf(M, K, seed=51), bf(N, K, scale=0.02, seed=52)
    w[:, 0] += (torch.arange(N, device=DEV) * 1e-3).to(DT)        # asymmetric: a transposed / shifted tile is an O(1) error
    ref = _gemm_ref(x, w, O.EPI_NONE, None, None)
    tol = 2 ** -7 * ref.abs().max().item() + 1e-3
    for cfg, sched in O.GEMM_CANDIDATES:
        y = O.gemm(x, w, config=O.gemm_config(cfg, sched))
        assert (y.float() - ref).abs().max().item() <= tol, (cfg, sched)
    y = O.gemm(x, w)                                                       # tuned choice
    assert (y.float() - ref).abs().max().item() <= tol
    big = bf(M, K + 64, seed=53)                                           # strided rows (a column slice of a wider activation)
    yb = O.gemm(big[:, :K], w)
    assert (yb.float() - _gemm_ref(big[:, :K], w, O.EPI_NONE, None, None)).abs().max().item() <= tol


# (8500 x 11008: 34 row-tiles x 86 column tiles of the SwiGLU product - the tile order switches to groups of 4 row-tiles there, the last group ragged)
@pytest.mark.parametrize("M,N,K", [(300, 512, 256), (768, 11008, 4096), (577, 4096, 1024), (96, 13824, 5120), (8500, 11008, 256)])
def test_gemm_epilogues(M, N, K):
    O = ops()
    x = bf(M, K, seed=54)
    bias, resid = bf(N, seed=56), bf(M, N, seed=57)
    for epi in (O.EPI_BIAS, O.EPI_BIAS_QUICK_GELU, O.EPI_BIAS_GELU, O.EPI_BIAS_RESID, O.EPI_SWIGLU):
        w = bf(2 * N if epi == O.EPI_SWIGLU else N, K, scale=0.03, seed=55)
        ref = _gemm_ref(x, w, epi, bias, resid)
        tol = 2 ** -6 * ref.abs().max().item() + 2e-3
        for cfg in range(1, 17):
            if epi == O.EPI_SWIGLU and cfg in (5, 6, 7, 9, 13, 14, 16): # 192-column tiles do not tile F % 128 features; one 32-column block per wave has no gate / up pair
                continue
            y = O.gemm(x, w, bias=bias, resid=resid, epi=epi, config=O.gemm_config(cfg))
            assert y.shape == (M, N) and (y.float() - ref).abs().max().item() <= tol, (epi, cfg)


def test_gemm_repeated_launches_leave_the_arrival_counters_clean():
    """Stream-K tiles are finished through arrival counters that every launch must leave at zero: many back-to-back launches
    of split shapes on one workspace, results identical every time."""
    O = ops()
    x, w = bf(768, 4096, seed=58), bf(4096, 4096, scale=0.02, seed=59)
    first = O.gemm(x, w, config=1 + 32).clone()
    for i in range(50):
        y = O.gemm(x, w, config=(1, 3, 4)[i % 3] + 32)
        assert (y.float() - first.float()).abs().max().item() <= 2 ** -7 * first.float().abs().max().item() + 1e-3
    assert torch.equal(O.gemm(x, w, config=1 + 32), first)


def test_gemm_small_launch_slabs_do_not_poison_a_large_launch_counters():
    """Round-3 regression (a hang found by tests/test_full_depth_gpu.py): the partial slabs used to start right behind the Mt x Nt
    counters of the CURRENT launch, so the fp32 slabs of a launch with few tiles lay where a later launch with many tiles reads
    its arrival counters (a negative float pattern = a finisher that spins forever; a positive one = partial sums added before
    they were written).  The counter region is fixed now: a stream-K launch with ONE tile row, then launches with thousands of
    tiles in every schedule, on one workspace."""
    O = ops()
    xs, w = bf(96, 4096, seed=60) * 50, bf(4096, 4096, scale=0.5, seed=61)             # large-magnitude slabs, both signs
    for cfg in (8, 2, 1):
        O.gemm(xs, w, config=cfg + 32)                                                     # stream-K only: every workgroup publishes a slab
    xb = bf(40000, 4096, seed=62)
    ref = _gemm_ref(xb, w, O.EPI_NONE, None, None)
    tol = 2 ** -7 * ref.abs().max().item() + 1e-3
    for cfg, sched in ((2, 0), (2, 2), (7, 0), (1, 2), (3, 0)):                          # up to 313 x 32 = 10,016 tiles
        y = O.gemm(xb, w, config=O.gemm_config(cfg, sched))
        assert (y.float() - ref).abs().max().item() <= tol, (cfg, sched)
    import ctypes
    from llava_align_amd import _lib
    lib = _lib.load_lib()
    lib.vdd_gemm_workspace_bytes.restype = ctypes.c_int64
    assert lib.vdd_gemm_workspace_bytes(96, 4096) == lib.vdd_gemm_workspace_bytes(40000, 4096)     # one layout for every shape


# row buckets of the prologue template (2 / 4 / 8 / 16) x block plan (two 4-wave blocks per CU up to 8 rows of 4096 / 6 of 5120, one 8-wave
# block beyond) x ragged K batches (5120 over 8 waves: 640 = 2 batches + 128)
@pytest.mark.parametrize("M,d,F", [(2, 4096, 11008), (3, 5120, 13824), (16, 4096, 11008), (5, 256, 512), (8, 4096, 11008), (9, 4096, 11008),
                                   (12, 4096, 11008), (7, 5120, 13824), (14, 5120, 13824), (1, 4096, 11008), (4, 8192, 1024)])
def test_norm_fused_small_m_projections(M, d, F):
    """The one-question decoder layer without RMSNorm launches: linear_resid_ss (projection + residual add + per-block sums of
    squares), linear_normed / swiglu_linear_normed (normalise-on-load) against rmsnorm + the plain projections.  Same bf16 rounding
    points; the only licence is the summation order of the fp32 sum of squares (one ulp of rstd)."""
    O = ops()
    x, resid = bf(M, d, seed=70), bf(M, d, seed=71)
    wo, ln = bf(d, d, scale=0.02, seed=72), (1 + 0.1 * bf(d, seed=73).float()).to(DT)
    h, ss = O.linear_resid_ss(x, wo, resid)
    want_h = O.skinny_gemm(x, wo, resid=resid)
    assert torch.equal(h, want_h)                                            # same kernel arithmetic as the unfused projection
    want_ss = (h.float() ** 2).view(M, d // 16, 16).sum(-1)
    assert torch.allclose(ss, want_ss, rtol=1e-5, atol=1e-6) and ss.shape == (M, d // 16)
    a = O.rmsnorm(h, ln, 1e-5)                                                # reference path: stand-alone norm, then the projections
    for w_next, bias in ((bf(3 * d, d, scale=0.02, seed=74), None), (bf(1000, d, scale=0.02, seed=75), bf(1000, seed=76))):
        got = O.linear_normed(h, ss, ln, 1e-5, w_next, bias=bias)
        want = O.linear(a, w_next, bias=bias)
        assert (got.float() - want.float()).abs().max().item() <= 2 ** -6 * want.float().abs().max().item()
        assert (got != want).float().mean().item() <= 0.02                   # bit-identical but for the odd rstd ulp
    wgu = bf(2 * F, d, scale=0.02, seed=77)
    got = O.swiglu_linear_normed(h, ss, ln, 1e-5, wgu)
    want = O.swiglu_linear(a, wgu)
    assert got.shape == (M, F) and (got.float() - want.float()).abs().max().item() <= 2 ** -6 * want.float().abs().max().item() + 1e-3
    assert (got != want).float().mean().item() <= 0.02


def test_gemm_rejects_what_it_cannot_do():
    O = ops()
    with pytest.raises(ValueError):
        O.gemm(bf(16, 200), bf(64, 200))                   # K % 128
    with pytest.raises(ValueError):
        O.gemm(bf(16, 256), bf(66, 256))                   # N % 4


def test_vit_glue_kernels():
    O = ops()
    n, S, P, H, D = 3, 56, 14, 2, 64
    G, T, w = S // P, (S // P) ** 2 + 1, H * D
    for dt in (torch.float32, torch.float16, DT):
        img = torch.randn(n, 3, S, S, device=DEV).to(dt)
        got = O.vit_im2col(img, P, 640, dtype=DT)
        want = img.to(DT).view(n, 3, G, P, G, P).permute(0, 2, 4, 1, 3, 5).reshape(n * G * G, 3 * P * P)
        assert torch.equal(got[:, : 3 * P * P], want) and not got[:, 3 * P * P:].any()
    emb, cls, pos = bf(n * (T - 1), w, seed=71), bf(w, seed=72), bf(T, w, seed=73)
    h = O.vit_assemble(emb, cls, pos, n, T)
    want = (torch.cat([cls.view(1, 1, w).expand(n, 1, w), emb.view(n, T - 1, w)], 1).float() + pos.float()[None]).to(DT)
    assert torch.equal(h.view(n, T, w), want)
    qkv = bf(n * T, 3 * w, seed=74)
    kc, vc = torch.zeros(n + 1, H, T + 3, D, dtype=DT, device=DEV), torch.zeros(n + 1, H, T + 3, D, dtype=DT, device=DEV)
    q = O.vit_qkv_split(qkv, kc, vc, n, T, H, D)
    v5 = qkv.view(n, T, 3, H, D)
    assert torch.equal(q.view(n, T, H, D), v5[:, :, 0])
    assert torch.equal(kc[:n, :, :T], v5[:, :, 1].permute(0, 2, 1, 3)) and torch.equal(vc[:n, :, :T], v5[:, :, 2].permute(0, 2, 1, 3))


def test_rmsnorm_sums_split_k_slabs():
    O = ops()
    M, d = 40, 4096
    x, w = bf(M, d, seed=61), bf(d, seed=62) * 0.1 + 1
    slabs = torch.randn(4, M, d, device=DEV) * 0.5
    ro = torch.empty_like(x)
    y = O.rmsnorm(x, w, 1e-5, delta=slabs, resid_out=ro)
    h = (x.float() + slabs.sum(0).to(DT).float()).to(DT)
    assert (ro.float() - h.float()).abs().max().item() <= 0.04          # fp32 summation order may move one bf16 ulp
    ref = (h.float() * torch.rsqrt(h.float().pow(2).mean(-1, keepdim=True) + 1e-5)).to(DT).float() * w.float()
    assert torch.allclose(y.float(), ref, rtol=3e-2, atol=2e-2)


@pytest.mark.parametrize("M,N,K", [(40, 4096, 4096), (64, 4096, 11008), (128, 5120, 13824), (33, 520, 512), (100, 4096, 4096),
                                   (12, 5120, 13824), (24, 5120, 5120), (4, 5120, 13824), (12, 4096, 11008), (200, 4096, 4096), (256, 4096, 11008)])     # 13B widths from 8 rows: slabs
def test_gemm_split_k_slabs_feed_the_rmsnorm(M, N, K):
    """Schedule 3 of vdd_gemm: fp32 slabs [S, M, N], one (tile, K part) per workgroup, no fix-up; their sum is the product, and
    rmsnorm(delta=slabs) == rmsnorm(delta=rounded product) up to the summation order."""
    O = ops()
    x, w = bf(M, K, seed=91), bf(N, K, scale=0.02, seed=92)
    ref = x.float() @ w.float().t()
    for S in sorted(s_ for s_ in {1, 2, 5, O.slab_splits(M, N, K) or 3, min(16, K // 128)} if s_ <= K // 128):
        sl = O.gemm_slabs(x, w, S)
        assert sl.shape == (S, M, N) and sl.dtype == torch.float32
        assert (sl.sum(0) - ref).abs().max().item() <= 1e-3 * ref.abs().max().item() + 1e-3, S
    res, lnw = bf(M, N, seed=93), bf(N, seed=94) * 0.1 + 1
    S = O.slab_splits(M, N, K) or 4
    r1, r2 = torch.empty_like(res), torch.empty_like(res)
    y1 = O.rmsnorm(res, lnw, 1e-5, delta=O.gemm_slabs(x, w, S), resid_out=r1)
    y2 = O.rmsnorm(res, lnw, 1e-5, delta=O.gemm(x, w), resid_out=r2)
    assert (r1.float() - r2.float()).abs().max().item() <= 2 ** -7 * r2.float().abs().max().item()
    assert (y1.float() - y2.float()).abs().max().item() <= 2 ** -6 * y2.float().abs().max().item()
    # the engine's entry picks ONE of the eligible forms by measurement (ops._pick_form); whichever it is, the norm behind it gives the same
    # residual stream and output to rounding, and the pick is recorded
    out = O.linear_to_norm(x, w)
    r3 = torch.empty_like(res)
    y3 = O.rmsnorm(res, lnw, 1e-5, delta=out, resid_out=r3)
    assert (r3.float() - r2.float()).abs().max().item() <= 2 ** -7 * r2.float().abs().max().item()
    assert (y3.float() - y2.float()).abs().max().item() <= 2 ** -6 * y2.float().abs().max().item()
    eligible = {"gemm"} | ({"skinny"} if O._skinny_serves(M, K) else set()) | ({"slabs"} if (M <= 256 and N <= 8192 and K % 256 == 0 and O.slab_splits(M, N, K)) else set())
    if len(eligible) > 1:
        pick = O._form_choice[O._form_key("to_norm", M, N, K, O._MODEL_DT[DT])]
        assert pick in eligible and (out.dim() == 3) == (pick == "slabs")
    for form in sorted(eligible):                                    # every eligible form on request (tests / probes: ops.FORCE_FORM)
        O.FORCE_FORM["to_norm"] = form
        try:
            o_ = O.linear_to_norm(x, w)
        finally:
            O.FORCE_FORM.pop("to_norm")
        assert (o_.dim() == 3) == (form == "slabs")
        yf = O.rmsnorm(res, lnw, 1e-5, delta=o_, resid_out=torch.empty_like(res))
        assert (yf.float() - y2.float()).abs().max().item() <= 2 ** -6 * y2.float().abs().max().item(), form


def test_attention_probs_materialises_the_causal_map():
    """vdd_attention_probs: [H, Tq, Tk] softmax of one sequence whose keys sit in [prefix slot | own slot] (GQA included), against torch."""
    O = ops()
    H, Hkv, D, plen, Ts = 8, 4, 128, 37, 21
    T = plen + Ts
    kp, ko = bf(3, Hkv, 64, D, seed=101), bf(5, Hkv, 32, D, seed=102)
    q = bf(T + 4, H * D, seed=103)
    got = O.attention_probs(q, ko, (2, T, 0, 4, 1, plen), H, Hkv, D, k_prefix=kp)             # rows 2 .. 2 + T of q; own slot 4, prefix slot 1
    K = torch.cat([kp[1, :, :plen], ko[4, :, :Ts]], 1).float().repeat_interleave(H // Hkv, 0)   # [H, T, D]
    qq = q[2:2 + T].view(T, H, D).float().permute(1, 0, 2)
    s_ = (qq @ K.transpose(1, 2)) * D ** -0.5
    mask = torch.ones(T, T, dtype=torch.bool, device=DEV).tril()
    want = s_.masked_fill(~mask, -float("inf")).softmax(-1)
    assert got.shape == (H, T, T) and got.dtype == DT
    assert (got.float() - want).abs().max().item() <= 2 ** -8 + 1e-3 and not got.float().triu(1).any()
    assert torch.allclose(got.float().sum(-1), torch.ones(H, T, device=DEV), atol=2e-2)
    own_only = O.attention_probs(q, ko, (0, 9, 0, 3, 0, 0), H, Hkv, D, k_prefix=kp)            # no prefix at all
    K2 = ko[3, :, :9].float().repeat_interleave(H // Hkv, 0)
    w2 = ((q[:9].view(9, H, D).float().permute(1, 0, 2) @ K2.transpose(1, 2)) * D ** -0.5).masked_fill(~mask[:9, :9], -float("inf")).softmax(-1)
    assert (own_only.float() - w2).abs().max().item() <= 2 ** -8 + 1e-3
