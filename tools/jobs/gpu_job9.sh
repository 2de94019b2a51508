#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2_job9_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r2_job9_tests.txt
(python tools/kbench.py --reps 300 --rounds 5; python tools/kbench.py --reps 300 --rounds 5 --streams 3; python tools/kbench_cfg.py) 2>&1 | grep -v amdgpu.ids > gpurun_out/r2_job9_kbench.txt
tail -3 gpurun_out/r2_job9_tests.txt; cat gpurun_out/r2_job9_kbench.txt
