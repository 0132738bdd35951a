#!/bin/bash
# One GPU call: A/B of the round-5 fusions, the 17 - 64-row curves.  Results under gpurun_out/.
mkdir -p gpurun_out
python tools/ab_fusions.py > gpurun_out/r05_ab_fusions.jsonl 2> gpurun_out/r05_ab_fusions.err
python tools/few_row_curve.py 6 8 11 16 21 > gpurun_out/r05_few_row_curve.jsonl 2> gpurun_out/r05_few_row_curve.err
VDD_MODEL=llava-1.5-13b python tools/few_row_curve.py 11 >> gpurun_out/r05_few_row_curve.jsonl 2>> gpurun_out/r05_few_row_curve.err
python tools/step_curve.py 1 9 12 16 24 32 48 64 96 128 > gpurun_out/r05_step_curve.jsonl 2> gpurun_out/r05_step_curve.err
tail -n 3 gpurun_out/r05_ab_fusions.jsonl gpurun_out/r05_few_row_curve.jsonl gpurun_out/r05_step_curve.jsonl
