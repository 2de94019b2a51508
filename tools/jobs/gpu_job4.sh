#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2_job4_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r2_job4_tests.txt
bash tools/kbench_variants.sh --reps 300 --rounds 5 > gpurun_out/r2_job4_variants.txt 2>&1
tail -5 gpurun_out/r2_job4_tests.txt; cat gpurun_out/r2_job4_variants.txt
