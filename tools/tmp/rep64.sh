cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for i in 1 2 3 4; do
  VDD_GEMM_DEFAULTS=off VDD_GEMM_CHOICES=gpurun_out/ch64_$i.json python tools/step_curve.py 32 2>&1 | grep questions
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/ch64_*.json')):
    d=json.load(open(f)); sec=list(d.values())[0]
    print(f, {k:v for k,v in sorted(sec.items()) if k.split(',')[0] in ('64','32')})
P
