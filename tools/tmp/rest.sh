cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export VDD_GEMM_DEFAULTS=off VDD_GEMM_CHOICES=$PWD/gpurun_out/choices_new.json
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -8
