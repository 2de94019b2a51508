#!/bin/bash
# first GPU job of round 3: parity of the new DCT/mel code + A/B against the round-2 library
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/a_tests.txt 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/a_tests.txt
bash tools/kbench_variants.sh > gpurun_out/a_kbench.txt 2>&1; cat gpurun_out/a_kbench.txt
bash tools/kbench_variants.sh --streams 3 > gpurun_out/a_kbench3.txt 2>&1; cat gpurun_out/a_kbench3.txt
