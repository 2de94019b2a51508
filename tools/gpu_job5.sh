#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2_job5_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r2_job5_tests.txt
python tools/kbench.py --reps 300 --rounds 5 2>&1 | grep -v amdgpu.ids > gpurun_out/r2_job5_kbench.txt
python tools/kbench.py --reps 300 --rounds 5 --streams 3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2_job5_kbench.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r2_job5_bench.json 2> gpurun_out/r2_job5_bench.err
tail -12 gpurun_out/r2_job5_tests.txt; cat gpurun_out/r2_job5_kbench.txt; cat gpurun_out/r2_job5_bench.json; tail -3 gpurun_out/r2_job5_bench.err; cat gpurun_out/parity_measured.json
