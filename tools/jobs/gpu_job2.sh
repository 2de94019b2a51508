#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
bash tools/kbench_variants.sh --reps 300 --rounds 5 > gpurun_out/r2_job2_variants.txt 2>&1
python -m pytest tests/test_gpu_distributed_nccl.py -m gpu -x -q > gpurun_out/r2_job2_nccl.txt 2>&1; echo "rc=$?" >> gpurun_out/r2_job2_nccl.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r2_job2_bench_a.json 2> gpurun_out/r2_job2_bench_a.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r2_job2_bench_b.json 2> gpurun_out/r2_job2_bench_b.err
cat gpurun_out/r2_job2_variants.txt; tail -5 gpurun_out/r2_job2_nccl.txt; cat gpurun_out/r2_job2_bench_a.json gpurun_out/r2_job2_bench_b.json; tail -3 gpurun_out/r2_job2_bench_a.err
