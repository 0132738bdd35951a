#!/bin/bash
# Round profile set (run on the GPU box through gpurun from the repo root): writes everything under gpurun_out/prof/.
#   1. rocprofv3 --kernel-trace --stats of `python bench.py --no-baselines` (the summary the bench line's numbers come from)
#   2. PMC FETCH_SIZE / WRITE_SIZE of the fused sampling kernel at the roofline shape (SEPARATE passes, counters only)
#   3. PMC MfmaUtil of the hand-written GEMM at decode / prefill sizes, the flash attention and the MFMA prefix pass
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export TMPDIR=/tmp
OUT=gpurun_out/prof
mkdir -p $OUT
# hard limits on every rocprofv3 run: when the traced process aborts, rocprofv3 keeps waiting for it (one such run cost 36 GPU-minutes)
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --no-baselines > $OUT/bench_line.json 2> $OUT/bench.err
[ -s $OUT/bench_line.json ] || { echo "bench.py under rocprofv3 produced no line (see $OUT/bench.err)"; exit 1; }
CSV=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python tools/trace_median.py "$CSV" 45 > $OUT/kernel_medians.txt
STATS=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
cp "$STATS" $OUT/kernel_stats.csv 2>/dev/null
python - "$CSV" > $OUT/fused_kernel_by_batch.txt <<'PY'
import csv, sys, collections, statistics
rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "vdd_contrast_sample_kernel" in r["Kernel_Name"]:
        rows[int(r["Grid_Size_X"]) // 512].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for b, v in sorted(rows.items()):
    print(f"B={b:5d} n={len(v):5d} mean={statistics.mean(v):8.2f} us median={statistics.median(v):8.2f} us")
# bench.py launches the roofline shape first (10 warm-up + 100 timed, beta = 0.1), then the beta = 1e-6 extra point (10 + 50) at the same grid
v = rows.get(4096, [])
if len(v) >= 170:
    print(f"B= 4096 roofline leg (launches 11-110, the timed ones of `roofline`): mean={statistics.mean(v[10:110]):8.2f} us")
    print(f"B= 4096 beta=1e-6 extra point (launches 121-170): mean={statistics.mean(v[120:170]):8.2f} us")
PY
rm -rf $OUT/trace
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 400 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o k -- python tools/kernel_sweep.py --only-canonical > /dev/null 2> $OUT/pmc_$C.err
  F=$(find $OUT/pmc_$C -name '*counter_collection.csv' | head -1)
  python - "$F" $C >> $OUT/pmc_fused_kernel.txt <<'PY'
import csv, sys, statistics
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "vdd_contrast_sample_kernel" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2] and int(r["Grid_Size"]) == 4096 * 512]
print(f"{sys.argv[2]:11s} n={len(v)} mean {statistics.mean(v):.1f} KB per launch (min {min(v):.1f}, max {max(v):.1f})" if v else f"{sys.argv[2]}: no rows")
PY
  rm -rf $OUT/pmc_$C
done
timeout -s KILL 600 rocprofv3 --pmc MfmaUtil --output-format csv -d $OUT/pmc_mfma -o k -- python tools/mfma_util_probe.py > /dev/null 2> $OUT/pmc_mfma.err
F=$(find $OUT/pmc_mfma -name '*counter_collection.csv' | head -1)
python - "$F" > $OUT/pmc_mfma_util.txt <<'PY'
import csv, sys, collections, statistics, re
rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == "MfmaUtil":
        n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:70]
        rows[(n, r["Grid_Size"])].append(float(r["Counter_Value"]))
for (n, g), v in sorted(rows.items(), key=lambda kv: -statistics.mean(kv[1])):
    print(f"MfmaUtil {statistics.mean(v):6.1f} %  n={len(v):4d}  grid={g:>10s}  {n}")
PY
rm -rf $OUT/pmc_mfma
ls -la $OUT
