#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/b_tests.txt 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/b_tests.txt
bash tools/kbench_variants.sh > gpurun_out/b_kbench.txt 2>&1; cat gpurun_out/b_kbench.txt
python tools/diag_nfilt64.py > gpurun_out/b_diag64.txt 2>&1; cat gpurun_out/b_diag64.txt
