"""Per-kernel median/mean/count from a rocprofv3 kernel trace CSV (median = the decode-step launch when decode calls dominate)."""
import csv, sys, collections, statistics, re
rows = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:60]
        rows[(n, r["Grid_Size_X"] + "x" + r["Grid_Size_Y"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
agg = sorted(((sum(v), k, v) for k, v in rows.items()), reverse=True)
tot = sum(a[0] for a in agg)
for s, (n, g), v in agg[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{s / 1e3:9.2f} ms {100 * s / tot:5.1f}% n={len(v):6d} med={statistics.median(v):8.1f} us grid={g:12s} {n}")
