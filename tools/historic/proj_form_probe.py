"""Wide projections (qkv, gate/up + SwiGLU, lm_head) of LLaVA-1.5-7B / 13B and Qwen-VL-7B at 4 - 40 rows: the weight-streaming kernels
(16 columns per block up to 16 rows, 32 above) against the MFMA GEMM, us per launch on rotating weights - where ops.skinny_rows() should
switch.  Record: profiles/r05_proj_form_probe.jsonl."""
import sys, json, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llava_align_amd import ops
dev = "cuda"
def t(fn, n=20):
    for i in range(4): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / n, 1)
shapes = [("13b_qkv", 15360, 5120, 0), ("13b_gu", 27648, 5120, 1), ("13b_lm_head", 32000, 5120, 0), ("7b_qkv", 12288, 4096, 0), ("7b_gu", 22016, 4096, 1), ("7b_lm_head", 32000, 4096, 0), ("qwen_lm_head", 151936, 4096, 0)]
for name, N, K, sw in shapes:
    nW = max(2, int(7e8 / (N * K * 2)))
    ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(nW)]
    for M in (4, 6, 8, 10, 12, 16, 17, 20, 24, 28, 32, 40):
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        if sw:
            out = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
            sk = t(lambda i: ops._lib.check(ops._lib_ready().vdd_skinny_swiglu(x.data_ptr(), ws[i % nW].data_ptr(), out.data_ptr(), M, N // 2, K, x.stride(0), 2, torch.cuda.current_stream().cuda_stream)))
            gm = t(lambda i: ops.gemm(x, ws[i % nW], epi=ops.EPI_SWIGLU))
        else:
            sk = t(lambda i: ops.skinny_gemm(x, ws[i % nW]))
            gm = t(lambda i: ops.gemm(x, ws[i % nW]))
        print(json.dumps({"op": name, "M": M, "skinny": sk, "gemm": gm}), flush=True)
    del ws
