#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "1536 or model or golden" > gpurun_out/r2_job22_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r2_job22_tests.txt
(DSP_F1536_NOGAP=1 python tools/kbench_cfg.py; python tools/kbench_cfg.py; DSP_F1536_NOGAP=1 python tools/kbench_cfg.py --batch 2048; python tools/kbench_cfg.py --batch 2048) 2>&1 | grep -v amdgpu.ids > gpurun_out/r2_job22_kbench.txt
tail -3 gpurun_out/r2_job22_tests.txt; cat gpurun_out/r2_job22_kbench.txt
