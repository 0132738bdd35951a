"""Where a launch of the 17 - 64-row slab projections (vdd_skinny_slab.hip) spends its time: per-wave phase stamps (X staged, main
loop done, team barrier passed, slabs summed, epilogue issued, done) of ONE launch per projection of LLaVA-1.5-7B.  Needs a library
built with -DVDD_PROBE_BUILD (exports vdd_dbg_slab_timeline_bf16).  Record: profiles/r05_slab_timeline.jsonl."""
import sys, os, torch, ctypes as C, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))                       # lost_ops.py, the lab tests
import lost_ops as O
lib = O._lib_ready()                      # build with `python lost_ops.py --probe` (VDD_PROBE_BUILD: the timeline stamps)
NW = 4   # SLAB_NW of vdd_skinny_slab.hip
dbg = torch.zeros(512 * NW * 8, dtype=torch.int64, device="cuda")
d, F = 4096, 11008
for M in (18, 64):
  for name, N, K, sw in (("qkv", 3 * d, d, False), ("o", d, d, False), ("gu", F, d, True), ("down", d, F, False)):
    x = torch.randn(M, K, device="cuda").bfloat16()
    ws = [(torch.randn((2 if sw else 1) * N, K, device="cuda") * 0.02).bfloat16() for _ in range(3)]
    for i in range(3): O.slab_linear(x, ws[i], swiglu=sw)
    torch.cuda.synchronize()
    dbg.zero_(); lib.vdd_dbg_slab_timeline_bf16(C.c_void_p(dbg.data_ptr()))
    O.slab_linear(x, ws[0], swiglu=sw); torch.cuda.synchronize()
    lib.vdd_dbg_slab_timeline_bf16(C.c_void_p(0))
    t = dbg.view(-1, 8).cpu()
    t = t[t[:, 0] > 0]
    base = t[:, 0].min()
    us = lambda c: ((t[:, c] - base).float() / 100.0)
    act = t[:, 7] > 0
    fz = t[:, 5] > 0
    rec = {"M": M, "op": name, "finishers": int(fz.sum()), "fin_ticket_to_sums_med": round(((t[fz, 5] - t[fz, 3]).float() / 100).median().item(), 2), "fin_ticket_to_sums_max": round(((t[fz, 5] - t[fz, 3]).float() / 100).max().item(), 2),
           "fin_sums_to_epi_med": round(((t[fz, 6] - t[fz, 5]).float() / 100).median().item(), 2), "fin_sums_to_epi_max": round(((t[fz, 6] - t[fz, 5]).float() / 100).max().item(), 2),
           "fin_epi_to_done_med": round(((t[fz, 4] - t[fz, 6]).float() / 100).median().item(), 2), "fin_epi_to_done_max": round(((t[fz, 4] - t[fz, 6]).float() / 100).max().item(), 2),
           "start_max": round(us(0).max().item(), 2), "staged_med": round(us(1).median().item(), 2), "staged_max": round(us(1).max().item(), 2),
           "loop_end_med": round(us(2)[act].median().item(), 2), "loop_end_max": round(us(2)[act].max().item(), 2),
           "ticket_med": round(us(3)[act].median().item(), 2), "ticket_max": round(us(3)[act].max().item(), 2),
           "done_med": round(us(4)[act].median().item(), 2), "done_max": round(us(4)[act].max().item(), 2)}
    print(json.dumps(rec), flush=True)
