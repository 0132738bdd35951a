"""The 17 - 64-row weight-streaming projections with K cut over workgroups (vdd_skinny_slab.hip, laboratory code since round 6) against fp32 references of
the same ops, in both storage types: plain / residual + sums of squares / normalise-on-staging / SwiGLU, ragged shapes, the ticket
protocol (repeated launches, bit-identical results, tickets left zero)."""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.probe]
DEV = "cuda:0"
DT = torch.bfloat16


@pytest.fixture(autouse=True, params=[torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def storage_dtype(request):
    global DT
    DT = request.param
    yield
    DT = torch.bfloat16


def ops():
    import lost_ops as O          # the slab entries + (through its __getattr__) the product's ops
    return O


def rt(*shape, scale=1.0, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, device=DEV, generator=g) * scale).to(DT)


def rnd(t):
    return t.to(DT).float()


def close(got, want, rel=2 ** -6):
    return (got.float() - want).abs().max().item() <= rel * want.abs().max().item() + 1e-6


# 7B / 13B widths (qkv, o, down, lm_head incl. a vocabulary that is not a multiple of 16) and small ragged shapes; M over the four
# M-tile counts and both sides of each boundary
SHAPES = [(17, 12288, 4096), (32, 4096, 4096), (33, 4096, 11008), (64, 12288, 4096), (64, 4096, 11008), (48, 32003, 4096), (34, 15360, 5120),
          (34, 5120, 13824), (20, 1000, 512), (64, 48, 256), (5, 4096, 4096), (16, 4096, 11008), (40, 272, 1056)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_slab_linear_matches_fp32_reference(M, N, K):
    O = ops()
    x, w = rt(M, K, seed=1), rt(N, K, scale=0.02, seed=2)
    want = x.float() @ w.float().T
    got = O.slab_linear(x, w)
    assert got.shape == (M, N) and close(got, rnd(want))
    # one rounding of an fp32 sum whose order differs from torch's: at most the last bit of a 16-bit float, on a minority of entries
    resid = rt(M, N, seed=3)
    got_r, ss = O.slab_linear(x, w, resid=resid, want_ss=True)
    want_r = rnd(rnd(want) + resid.float())
    assert close(got_r, want_r)
    NT = (N + 15) // 16
    pad = torch.zeros(M, NT * 16, device=DEV)
    pad[:, :N] = got_r.float() ** 2
    assert ss.shape == (M, NT) and torch.allclose(ss, pad.view(M, NT, 16).sum(-1), rtol=1e-5, atol=1e-6)   # of the values it WROTE


@pytest.mark.parametrize("M,d,F", [(18, 4096, 11008), (64, 4096, 11008), (34, 5120, 13824), (24, 512, 1376), (40, 256, 288)])
def test_slab_layer_ops_equal_the_unfused_ops(M, d, F):
    """o-projection (+ residual, sums of squares) -> normalise-on-staging qkv / gate-up SwiGLU -> down: against rmsnorm + the plain slab
    projections (the same kernel arithmetic, so equal but for the odd ulp of rstd) and against fp32."""
    O = ops()
    x, resid = rt(M, d, seed=10), rt(M, d, seed=11)
    wo, ln = rt(d, d, scale=0.02, seed=12), (1 + 0.1 * rt(d, seed=13).float()).to(DT)
    h, ss = O.slab_linear(x, wo, resid=resid, want_ss=True)
    assert close(h, rnd(rnd(x.float() @ wo.float().T) + resid.float()))
    a = O.rmsnorm(h, ln, 1e-5)
    wq = rt(3 * d, d, scale=0.02, seed=14)
    got, want = O.slab_linear(h, wq, ss=ss, ln_w=ln, eps=1e-5), O.slab_linear(a, wq)
    assert close(got, want.float()) and (got != want).float().mean().item() <= 0.02
    wgu = rt(2 * F, d, scale=0.02, seed=15)
    act, act_want = O.slab_linear(h, wgu, ss=ss, ln_w=ln, eps=1e-5, swiglu=True), O.slab_linear(a, wgu, swiglu=True)
    assert act.shape == (M, F) and close(act, act_want.float()) and (act != act_want).float().mean().item() <= 0.02
    gu = a.float() @ wgu.float().T
    g, u = rnd(gu[:, :F]), rnd(gu[:, F:])
    ref = rnd(rnd(g / (1 + torch.exp(-g))) * u)
    assert (act_want.float() - ref).abs().max().item() <= 2 ** -5 * ref.abs().max().item() + 1e-3
    wd = rt(d, F, scale=0.02, seed=16)
    h2, ss2 = O.slab_linear(act, wd, resid=h, want_ss=True)
    assert close(h2, rnd(rnd(act.float() @ wd.float().T) + h.float()))
    assert torch.allclose(ss2.sum(-1), (h2.float() ** 2).sum(-1), rtol=1e-4)


def test_slab_repeated_launches_are_bit_identical_and_leave_the_tickets_zero():
    O = ops()
    M, N, K = 40, 4096, 11008
    x, w, resid = rt(M, K, seed=20), rt(N, K, scale=0.02, seed=21), rt(M, N, seed=22)
    ws = torch.zeros(64 << 20, dtype=torch.uint8, device=DEV)
    first = O.slab_linear(x, w, resid=resid, workspace=ws)
    for _ in range(20):
        again = O.slab_linear(x, w, resid=resid, workspace=ws)
        assert torch.equal(first, again)
    # the partial slabs of a smaller launch must not poison a larger one that follows on the same workspace (and vice versa)
    small = O.slab_linear(x[:18, :4096].contiguous(), w[:, :4096].contiguous(), workspace=ws)
    assert close(small, rnd(x[:18, :4096].float() @ w[:, :4096].float().T))
    assert torch.equal(first, O.slab_linear(x, w, resid=resid, workspace=ws))
    torch.cuda.synchronize()
    assert int(ws[:1 << 16].view(torch.int32).abs().sum().item()) == 0


def test_slab_rejects_what_it_cannot_do():
    O = ops()
    assert not O.slab_serves(65, 4096, 4096) and not O.slab_serves(32, 4096, 4100) and O.slab_serves(64, 4096, 4096)
    with pytest.raises(ValueError):
        O.slab_linear(rt(65, 4096), rt(64, 4096))
