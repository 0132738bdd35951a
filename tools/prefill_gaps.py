"""GPU-idle holes inside the prefill phases of a rocprofv3 kernel trace of bench.py (usage: prefill_gaps.py r_kernel_trace.csv)."""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:50]))
rows.sort()
mark = [i for i, r in enumerate(rows) if "decode_attn_own_merge" in r[2] or "decode_attn_combine" in r[2]]
regions = [(a, b) for a, b in zip(mark, mark[1:]) if b - a > 1500]
for a, b in regions[:8]:
    seg = rows[a + 1:b]
    st = next(i for i, r in enumerate(seg) if "layernorm" in r[2] or "flash" in r[2])
    lead = (seg[st][0] - rows[a][1]) / 1e6
    seg = seg[st:]
    span = (seg[-1][1] - seg[0][0]) / 1e6
    busy = sum(r[1] - r[0] for r in seg) / 1e6
    gaps = sorted(((seg[i + 1][0] - seg[i][1]) / 1e6, seg[i][2][:26], seg[i + 1][2][:26]) for i in range(len(seg) - 1))[-3:]
    print(f"idle before the prefill {lead:6.1f} ms | prefill span {span:7.1f} ms busy {busy:7.1f} ms | top holes {[(round(g[0], 1), g[1], g[2]) for g in gaps]}")
