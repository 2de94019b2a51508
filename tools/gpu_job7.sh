#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2_job7_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r2_job7_tests.txt
(echo "--- vad_sum_kernel"; python tools/kbench_vad.py; echo "--- DSP_VAD_NOSUM=1 (round-1 kernels)"; DSP_VAD_NOSUM=1 python tools/kbench_vad.py; python tools/kbench_vad_dense.py; DSP_VAD_NOSUM=1 python tools/kbench_vad_dense.py) 2>&1 | grep -v amdgpu.ids > gpurun_out/r2_job7_vad.txt
tail -4 gpurun_out/r2_job7_tests.txt; cat gpurun_out/r2_job7_vad.txt
