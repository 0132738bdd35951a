"""GPU timeline of ONE one-question prefill (generate(max_new_tokens=1)) from a rocprofv3 kernel trace of tools/b1_host_profile.py:
the last call's kernels by name, its busy time and its wall span.  usage: python tools/b1_prefill_trace.py <kernel_trace.csv> [n_calls]"""
import csv, sys, re, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:48]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r["Grid_Size_X"] + "x" + r["Grid_Size_Y"]))
rows.sort()
# calls are separated by the sampling kernel (one per generate with max_new_tokens=1)
ends = [i for i, r in enumerate(rows) if "vdd_contrast_sample" in r[2]]
lo, hi = ends[-2] + 1, ends[-1] + 1
call = rows[lo:hi]
busy = sum(e - s for s, e, _, _ in call) / 1e3
print(f"last call: {len(call)} launches, busy {busy / 1e3:.2f} ms, span {(call[-1][1] - call[0][0]) / 1e6:.2f} ms")
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, g in call:
    agg[(n, g)][0] += 1; agg[(n, g)][1] += (e - s) / 1e3
for (n, g), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{t / 1e3:7.2f} ms n={c:4d} avg={t / c:7.1f} us grid={g:12s} {n}")
