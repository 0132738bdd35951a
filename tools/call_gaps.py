"""GPU-idle holes inside ONE generate() call of the POPE-proper batch (768 questions, 2 new tokens): run under rocprofv3 --kernel-trace, then
`call_gaps.py analyse <kernel_trace.csv>` (NEW_TOKENS=64: the headline step; WHAT=config2_full | config5 | config4: the timed call of that bench.py leg).  The timed call sits between two marker launches (fills of a float64 tensor)."""
import csv, json, os, sys, time
if sys.argv[1:2] == ["analyse"]:
    rows = []
    for r in csv.DictReader(open(sys.argv[2])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:60],
                     int(r.get("Grid_Size_X", 0) or 0), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "FillFunctor<double>" in r[4]]
    a, b = marks[-2], marks[-1]
    seg = rows[a + 1:b]
    span, busy = (seg[-1][1] - seg[0][0]) / 1e6, sum(r[1] - r[0] for r in seg) / 1e6
    holes = sorted(((seg[i + 1][0] - seg[i][1]) / 1e3, seg[i][2][:40], seg[i + 1][2][:40], i, round((seg[i][1] - seg[0][0]) / 1e6, 1)) for i in range(len(seg) - 1))
    big = [h for h in holes if h[0] > 15]
    print(json.dumps({"kernels": len(seg), "lead_ms": round((seg[0][0] - rows[a][1]) / 1e6, 2), "tail_ms": round((rows[b][0] - seg[-1][1]) / 1e6, 2),
                      "span_ms": round(span, 1), "busy_ms": round(busy, 1), "idle_ms": round(span - busy, 1),
                      "holes_over_15us": len(big), "their_sum_ms": round(sum(h[0] for h in big) / 1e3, 1),
                      "small_holes_sum_ms": round(sum(h[0] for h in holes if h[0] <= 15) / 1e3, 1)}))
    for h in holes[-25:][::-1]:
        print(f"{h[0]:9.1f} us   after kernel #{h[3]:5d} at +{h[4]:7.1f} ms   {h[1]:42s} -> {h[2]}")
    for h in holes[-4:]:
        i = h[3]
        print("---- around the", round(h[0] / 1e3, 1), "ms hole:")
        for j in range(max(0, i - 4), min(len(seg), i + 6)):
            print(f"   #{j:5d} +{(seg[j][0] - seg[0][0]) / 1e6:8.2f} ms  {(seg[j][1] - seg[j][0]) / 1e3:8.1f} us  grid {seg[j][3]:9d}  {seg[j][2]}")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bench import pope_prompts
from llava_align_amd.engine import VddLlavaEngine
dev = "cuda:0"
what = os.environ.get("WHAT", "pope")
m = torch.empty(54321, device=dev, dtype=torch.float64)
if what != "pope":
    # a driver leg of bench.py (config2_full / config5 / config4): the markers go around ITS timed call
    def timed(fn, d):
        torch.cuda.synchronize(d)
        m.fill_(1.0); torch.cuda.synchronize(d)
        t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(d)
        dt = time.perf_counter() - t0
        m.fill_(2.0); torch.cuda.synchronize(d)
        return out, dt
    bench._timed = timed
    if what == "config4":
        res = bench.bench_config4(torch.device(dev))
    else:
        eng = VddLlavaEngine("llava-1.5-7b", device=dev, seed=0, use_graph=True, lm_head_gain=4.0)
        res = getattr(bench, "bench_" + what)(eng, torch.device(dev))
    print(json.dumps({k: v for k, v in res.items() if k in ("seconds", "items_per_s", "prefill_s", "decode_s", "front_end_and_host_s")}))
    sys.exit(0)
eng = VddLlavaEngine("llava-1.5-7b", device=dev, seed=0, use_graph=True, lm_head_gain=4.0)
ids, host_imgs = pope_prompts(128, seed=1234, vocab=eng.cfg.lm.vocab, image=eng.cfg.vision.image)
on_dev = {}
imgs = [on_dev.setdefault(id(im), im.to(dev).to(torch.bfloat16)) for im in host_imgs]
kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, seed=1, n_top=10, max_new_tokens=int(os.environ.get("NEW_TOKENS", "2")))
for _ in range(2):
    eng.generate(ids, **kw)
torch.cuda.synchronize()
m.fill_(1.0); torch.cuda.synchronize()
t0 = time.perf_counter()
eng.generate(ids, **kw)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
m.fill_(2.0); torch.cuda.synchronize()
print(json.dumps({"wall_s": round(dt, 4)}))
