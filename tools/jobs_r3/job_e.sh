#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/e_tests.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/e_tests.txt
bash tools/kbench_variants.sh > gpurun_out/e_kbench.txt 2>&1; cat gpurun_out/e_kbench.txt
bash tools/kbench_variants.sh --streams 3 > gpurun_out/e_kbench3.txt 2>&1; cat gpurun_out/e_kbench3.txt
