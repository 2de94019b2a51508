#!/bin/bash
# round-2 job 1: MFMA probe, GPU test suite on the new grid mapping, kbench A/B of the two grid modes
set -o pipefail
mkdir -p gpurun_out
./tools/probes/mfma_f32_probe.bin > gpurun_out/r2_mfma_probe.txt 2>&1 || echo "probe rc=$?" >> gpurun_out/r2_mfma_probe.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r2_job1_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r2_job1_tests.txt
for mode in 0 1; do
  for b in 512 1024 2048; do
    DSP_F512_GRID=$mode python tools/kbench.py --batch $b --reps 200 --rounds 5 >> gpurun_out/r2_job1_kbench.txt 2>&1
  done
  DSP_F512_GRID=$mode python tools/kbench.py --batch 1024 --streams 3 --reps 200 --rounds 5 >> gpurun_out/r2_job1_kbench.txt 2>&1
done
tail -3 gpurun_out/r2_job1_tests.txt; cat gpurun_out/r2_mfma_probe.txt gpurun_out/r2_job1_kbench.txt
