"""Every (macro tile, schedule) candidate of the MFMA GEMM with guard bands around the output and around the workspace (arrival
counters + stream-K partial slabs): the result matches an fp32 matmul + epilogue, no byte outside the buffers changes, and the
arrival counters are back at zero after each launch (the next launch on the stream relies on that).  The tuner picks among these
candidates by timing, so each of them has to be safe at every shape class: ragged M, N of one or many column tiles, every epilogue."""
import pytest
import torch
import torch.nn.functional as F

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


GUARD = 1 << 18


def _cases():
    from llava_align_amd import ops as E
    return [(66, 12288, 4096, E.EPI_NONE), (192, 4096, 4096, E.EPI_BIAS_RESID), (130, 22016, 4096, E.EPI_SWIGLU), (96, 4096, 11008, E.EPI_NONE),
            (577, 3072, 1024, E.EPI_BIAS), (577, 4096, 1024, E.EPI_BIAS_QUICK_GELU), (1154, 1024, 4096, E.EPI_BIAS_RESID), (577, 1024, 640, E.EPI_NONE),
            (300, 4096, 1024, E.EPI_BIAS_GELU), (1536, 4096, 4096, E.EPI_NONE), (2900, 5120, 5120, E.EPI_NONE)]


@pytest.mark.parametrize("case", range(11))
def test_every_gemm_candidate_stays_inside_its_buffers(case):
    from llava_align_amd import ops as E
    M, N, K, epi = _cases()[case]
    g = torch.Generator(device=DEV).manual_seed(case)
    x = (torch.randn(M, K, device=DEV, generator=g) * 0.5).to(DT)
    w = (torch.randn(N, K, device=DEV, generator=g) * 0.02).to(DT)
    No = N // 2 if epi == E.EPI_SWIGLU else N
    bias = (torch.randn(No, device=DEV, generator=g) * 0.1).to(DT) if epi in (E.EPI_BIAS, E.EPI_BIAS_QUICK_GELU, E.EPI_BIAS_GELU, E.EPI_BIAS_RESID) else None
    resid = (torch.randn(M, No, device=DEV, generator=g) * 0.5).to(DT) if epi == E.EPI_BIAS_RESID else None
    acc = x.float() @ w.float().t()
    if epi == E.EPI_SWIGLU:
        want = F.silu(acc[:, :No].to(DT).float()).to(DT).float() * acc[:, No:].to(DT).float()
    else:
        want = acc if bias is None else acc + bias.float()
        if epi == E.EPI_BIAS_QUICK_GELU:
            want = want * torch.sigmoid(1.702 * want)
        if epi == E.EPI_BIAS_GELU:
            want = F.gelu(want)
        if resid is not None:
            want = want.to(DT).float() + resid.float()
    need = E._gemm_workspace(x.device, M, No).numel()
    scale = want.abs().max().item()
    n = 0
    for c, sch in E.GEMM_CANDIDATES:
        if (epi == E.EPI_SWIGLU and c in (5, 6, 7, 9, 13, 14, 16)) or (c == 8 and M > 256):        # (9 - 15: the tuner tries them up to 256 / 64 / 32 rows; they are valid at any M)
            continue
        wsbuf = torch.full((need + 2 * GUARD,), 0x5A, dtype=torch.uint8, device=DEV)
        ws = wsbuf[GUARD:GUARD + need]
        ws.zero_()
        obuf = torch.full((M * No + 2 * GUARD,), -7.0, dtype=DT, device=DEV)
        out = obuf[GUARD:GUARD + M * No].view(M, No)
        for _ in range(2):
            E._gemm_call(x, w, out, bias, resid, M, No, K, epi, E.gemm_config(c, sch), ws)
        torch.cuda.synchronize()
        assert (wsbuf[:GUARD] == 0x5A).all() and (wsbuf[GUARD + need:] == 0x5A).all(), (c, sch, "workspace guard")
        assert (obuf[:GUARD] == -7.0).all() and (obuf[GUARD + M * No:] == -7.0).all(), (c, sch, "output guard")
        assert (ws[: 4 << 20] == 0).all(), (c, sch, "arrival counters not back at zero")
        assert (out.float() - want).abs().max().item() <= 0.03 * scale, (c, sch)
        n += 1
    assert n >= 15
