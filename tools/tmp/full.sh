cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export VDD_GEMM_DEFAULTS=off VDD_GEMM_CHOICES=$PWD/gpurun_out/choices_new.json
rm -f $VDD_GEMM_CHOICES
python bench.py > gpurun_out/bench_r04c.json 2> gpurun_out/bench_r04c.err; tail -c 600 gpurun_out/bench_r04c.err
python -c "
import json; d=json.load(open('gpurun_out/bench_r04c.json')); print({k:d[k] for k in ('value','ms_per_step','prefill_plus_first_token_s','hbm_peak_GB')}, d['decode_step'], d['pope_eos']['questions_per_s_per_gpu'], d['single_question']['tokens_per_s'], d['fp16']['tokens_per_s_per_gpu'])"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
