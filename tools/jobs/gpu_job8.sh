#!/bin/bash
mkdir -p gpurun_out
bash tools/kbench_variants.sh --reps 300 --rounds 5 > gpurun_out/r2_job8_variants.txt 2>&1
cat gpurun_out/r2_job8_variants.txt
