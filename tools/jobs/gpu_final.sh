#!/bin/bash
mkdir -p gpurun_out
bash tools/profile_r2.sh > gpurun_out/prof_r2_run.txt 2>&1
python bench.py > gpurun_out/r2_bench_default.json 2> gpurun_out/r2_bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r2_bench_driver.json 2> gpurun_out/r2_bench_driver.err
tail -3 gpurun_out/r2_bench_default.err; python - <<'PY'
import json
for f in ('gpurun_out/r2_bench_default.json','gpurun_out/r2_bench_driver.json'):
    j=json.load(open(f)); r=j['roofline']
    print(f, 'value %.4g ms %.4f kernel_ms %.4f frac %.4f overl %.4f compute %.3f'%(j['value'],j['ms_per_step'],r['kernel_ms'],r['frac'],r['kernel_ms_overlapped'],r['compute']['frac']), j['timing']['blocks_of_K_steps'])
    if 'other_paths' in j: print(j['other_paths']); print(j.get('cpu_baseline'))
PY
