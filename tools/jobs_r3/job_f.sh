#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_batch.py -m gpu -x -q -k "fused_delta" > gpurun_out/f_fused.txt 2>&1; echo "fused rc=$?"; tail -15 gpurun_out/f_fused.txt
