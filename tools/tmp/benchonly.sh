cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export VDD_GEMM_DEFAULTS=off VDD_GEMM_CHOICES=$PWD/gpurun_out/choices_bench_new.json
rm -f $VDD_GEMM_CHOICES
python bench.py > gpurun_out/bench_r04d.json 2> gpurun_out/bench_r04d.err
python -c "
import json; d=json.load(open('gpurun_out/bench_r04d.json')); print({k:d[k] for k in ('value','ms_per_step','prefill_plus_first_token_s','hbm_peak_GB')}, d['decode_step']['ms'], d['pope_eos']['questions_per_s_per_gpu'], d['single_question']['tokens_per_s'], d['fp16']['tokens_per_s_per_gpu'], d['fp16']['prefill_plus_first_token_s'])"
