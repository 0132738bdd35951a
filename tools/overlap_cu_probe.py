"""N1 (north_star: "the two branches share a KV cache on one HIP stream pair"): ONE designed attempt at running the two bounds of the
1,536-row decode step concurrently - the MFMA-bound projections of one half-batch on a persistent grid of P < 256 workgroups
(stream masked to P CUs) while the HBM-bound decode attention of the OTHER half-batch runs on the remaining CUs
(hipExtStreamCreateWithCUMask), halves swapping roles every phase.  Needs a library built with -DVDD_PROBE_BUILD
(VDD_PROBE_GEMM_WORKGROUPS sets the grid).  python tools/overlap_cu_probe.py --P 192 --pattern word
Prints per layer: serial (what the engine runs: 1,536-row GEMMs then 1,536-row attention), serial halves, and the anti-phase pair."""
import argparse, ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--P", type=int, default=192)
ap.add_argument("--pattern", default="word", choices=["word", "byte", "none"])
ap.add_argument("--groups", type=int, default=64, help="image groups of 6 questions per HALF (64 -> 768 rows per half)")
ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
os.environ["VDD_PROBE_GEMM_WORKGROUPS"] = str(a.P)
import torch
from llava_align_amd import ops
dev = "cuda:0"
H = Hkv = 32; D = 128
PER, PL, UPL, OWN = 6, 611, 36, 57
bf = lambda *s: torch.randn(*s, device=dev, dtype=torch.bfloat16)


def half(G):
    Q = G * PER; M = 2 * Q
    ko, vo = bf(M, Hkv, 128, D), bf(M, Hkv, 128, D)
    kp, vp = bf(G + 1, Hkv, 640, D), bf(G + 1, Hkv, 640, D)
    pf = torch.empty((vp.shape[0], vp.shape[1], 2 * vp.shape[2], vp.shape[3]), dtype=vp.dtype, device=dev)
    ops.prefix_fragments(kp, vp, pf, torch.tensor([PL] * G + [UPL], dtype=torch.int32, device=dev))
    rows, groups, members = [], [], []
    for g in range(G):
        groups.append([len(members), PER, g, PL])
        for i in range(PER):
            members.append(len(rows)); rows.append([len(rows), PL + OWN, g, PL])
    groups.append([len(members), Q, G, UPL])
    for i in range(Q):
        members.append(len(rows)); rows.append([len(rows), UPL + OWN, G, UPL])
    q = bf(M, H * D)
    i32 = lambda x: torch.tensor(x, dtype=torch.int32, device=dev)
    cpi = ops.prefix_chunks_per_item(groups, H)
    it = ops.prefix_work_items(groups, cpi)
    st = dict(M=M, q=q, ko=ko, vo=vo, kp=kp, vp=vp, pf=pf, rt=i32(rows), gt=i32(groups), mt=i32(members), itt=i32(it), n_it=len(it), cpi=cpi,
              ws=ops.attention_workspace(M, H, D, 640 + 128, dev), out=torch.empty_like(q),
              X=[bf(M, 4096), bf(M, 4096), bf(M, 4096), bf(M, 11008)])
    st["Y"] = [torch.empty(M, n, device=dev, dtype=torch.bfloat16) for n in (12288, 4096, 11008, 4096)]
    return st


W = [bf(n, k) * 0.02 for n, k in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008))]


def attn(s):
    ops.decode_attention_grouped(s["q"], s["ko"], s["vo"], s["kp"], s["vp"], s["rt"], s["gt"], s["mt"], s["itt"], s["n_it"], H, Hkv, D, PL, OWN,
                                 out=s["out"], workspace=s["ws"], prefix_frag=s["pf"], chunks_per_item=s["cpi"])


def gemms(s):
    ops.gemm(s["X"][0], W[0], out=s["Y"][0])
    ops.gemm(s["X"][1], W[1], out=s["Y"][1])
    ops.gemm(s["X"][2], W[2], epi=ops.EPI_SWIGLU, out=s["Y"][2])
    ops.gemm(s["X"][3], W[3], out=s["Y"][3])


def masked_stream(bits):
    hip = ctypes.CDLL("libamdhip64.so")
    n_words = 8
    arr = (ctypes.c_uint32 * n_words)(*[sum(1 << b for b in range(32) if bits[w * 32 + b]) for w in range(n_words)])
    sp = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(sp), ctypes.c_uint32(n_words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(sp.value)


n_cu = torch.cuda.get_device_properties(0).multi_processor_count
if a.pattern == "word":
    g_bits = [(b % 32) < a.P * 32 // n_cu for b in range(256)]
elif a.pattern == "byte":
    g_bits = [(b % 8) < a.P * 8 // n_cu for b in range(256)]
else:
    g_bits = [True] * 256
A, B, full = half(a.groups), half(a.groups), half(2 * a.groups)
if a.pattern == "none":
    sg, st_ = torch.cuda.Stream(), torch.cuda.Stream()
else:
    sg, st_ = masked_stream(g_bits), masked_stream([not x for x in g_bits])


def timed(fn, n=a.iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def pair():
    cur = torch.cuda.current_stream()
    sg.wait_stream(cur); st_.wait_stream(cur)
    for x, y in ((A, B), (B, A)):
        with torch.cuda.stream(sg):
            gemms(x)
        with torch.cuda.stream(st_):
            attn(y)
        sg.wait_stream(st_); st_.wait_stream(sg)
    cur.wait_stream(sg); cur.wait_stream(st_)


out = dict(P=a.P, pattern=a.pattern, rows_per_half=A["M"], gemm_mask_cus=sum(g_bits))
# tune / warm every shape on the streams they will run on (workspaces are per stream)
for s_ in (A, B, full):
    gemms(s_); attn(s_)
with torch.cuda.stream(sg):
    gemms(A); gemms(B)
with torch.cuda.stream(st_):
    attn(A); attn(B)
torch.cuda.synchronize()
out["serial_full_us"] = round(timed(lambda: (gemms(full), attn(full))), 1)
out["gemm_full_us"] = round(timed(lambda: gemms(full)), 1)
out["attn_full_us"] = round(timed(lambda: attn(full)), 1)
out["serial_halves_us"] = round(timed(lambda: (gemms(A), attn(A), gemms(B), attn(B))), 1)


def on(stream, fn):
    def f():
        cur = torch.cuda.current_stream()
        stream.wait_stream(cur)
        with torch.cuda.stream(stream):
            fn()
        cur.wait_stream(stream)
    return f


out["gemm_half_on_P_us"] = round(timed(on(sg, lambda: gemms(A))), 1)
out["attn_half_on_rest_us"] = round(timed(on(st_, lambda: attn(B))), 1)
out["anti_phase_pair_us"] = round(timed(pair), 1)
out["ratio_vs_serial_full"] = round(out["anti_phase_pair_us"] / out["serial_full_us"], 3)
print(json.dumps(out))
