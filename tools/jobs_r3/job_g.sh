#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/g_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/g_tests.txt
for s in 1 3; do
DSP_F512_NOFUSE=1 python tools/kbench.py --streams $s 2>&1 | grep -v amdgpu.ids | sed 's/^/nofuse /'
python tools/kbench.py --streams $s 2>&1 | grep -v amdgpu.ids | sed 's/^/fused  /'
done
python tools/kbench.py --batch 12500 --reps 20 2>&1 | grep -v amdgpu.ids | sed 's/^/fused 12500 /'
