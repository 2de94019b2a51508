#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/d_tests.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/d_tests.txt
DSP_FRONTEND_LIB=$PWD/dsp-speech-recognition_amd/lib/variants/stamps.so python tools/kbench.py > gpurun_out/d_stamps.txt 2>&1; cat gpurun_out/d_stamps.txt
bash tools/pmc_run.sh d_pmc > gpurun_out/d_pmc.txt 2>&1; cat gpurun_out/d_pmc.txt
