set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; export TMPDIR=/tmp
for NQ in 96 16; do
  O=gpurun_out/steptrace_$NQ; rm -rf $O; mkdir -p $O
  timeout -s KILL 500 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python tools/step_curve.py $NQ > $O/line.json 2> $O/err.txt
  CSV=$(find $O/trace -name '*kernel_trace.csv' | head -1)
  python tools/trace_median.py "$CSV" 30 > $O/medians.txt
  rm -rf $O/trace
  cat $O/line.json; head -30 $O/medians.txt | cut -c1-170
done
