#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/h_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/h_tests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/h_bench.json 2> gpurun_out/h_bench.err; echo "bench rc=$?"; tail -5 gpurun_out/h_bench.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/h_bench.json')); r=j['roofline']
print('value %.4g ms %.4f kernel_ms %.4f frac %.4f overl %.4f'%(j['value'],j['ms_per_step'],r['kernel_ms'],r['frac'],r['kernel_ms_overlapped']), r['compute'], r['mfcc_only_kernel'])
print(json.dumps(j.get('other_paths'), indent=1)); print(j.get('cpu_baseline'))
PY
