set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
python -m pytest tests/test_llm_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -3
python tools/gemm_m_curve.py 66 96 128 160 192 256 384 > gpurun_out/m_curve_stream.jsonl 2>&1; cat gpurun_out/m_curve_stream.jsonl
VDD_GEMM_DEFAULTS=off VDD_GEMM_CHOICES=off python tools/step_curve.py 32 48 64 96 128 > gpurun_out/step_stream.jsonl 2>&1; cat gpurun_out/step_stream.jsonl
