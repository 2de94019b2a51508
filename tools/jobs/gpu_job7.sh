#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2_job7_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r2_job7_tests.txt
python tools/kbench_vad.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2_job7_vad.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_job7_bench.json 2> gpurun_out/r2_job7_bench.err
tail -4 gpurun_out/r2_job7_tests.txt; cat gpurun_out/r2_job7_vad.txt; python -c "
import json; j=json.load(open('gpurun_out/r2_job7_bench.json')); print(j['other_paths']['configs3_vad_pipeline'])"
