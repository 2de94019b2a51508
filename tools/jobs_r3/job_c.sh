#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/c_tests.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/c_tests.txt
bash tools/kbench_variants.sh > gpurun_out/c_kbench.txt 2>&1; cat gpurun_out/c_kbench.txt
bash tools/kbench_variants.sh > gpurun_out/c_kbench_again.txt 2>&1; cat gpurun_out/c_kbench_again.txt
