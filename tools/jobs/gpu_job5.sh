#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2_job5_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r2_job5_tests.txt
python tools/kbench.py --reps 300 --rounds 5 2>&1 | grep -v amdgpu.ids > gpurun_out/r2_job5_kbench.txt
python tools/kbench.py --reps 300 --rounds 5 --streams 3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2_job5_kbench.txt
python tools/kbench_ragged.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2_job5_kbench.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_job5_bench.json 2> gpurun_out/r2_job5_bench.err
tail -4 gpurun_out/r2_job5_tests.txt; cat gpurun_out/r2_job5_kbench.txt; python - <<'PY'
import json
j=json.load(open('gpurun_out/r2_job5_bench.json'))
print('value',j['value'],'ms',j['ms_per_step'],'kernel_ms',j['roofline']['kernel_ms'],'frac',j['roofline']['frac'])
print(j['other_paths'])
PY
