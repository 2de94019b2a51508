#!/bin/bash
# Round-4 rocprofv3 evidence (run on the GPU box from the repo root: gpurun -- "bash tools/profile_r4.sh").
# Kernel-trace / stats passes and every --pmc pass are separate runs (never combined with other trace domains).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_r4
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B1="$R/bench.py --steps 200 --warmup 20 --streams 1 --no-cpu-baseline --no-extras"
B3="$R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras"
KB="$R/tools/kbench.py --reps 10 --rounds 1"
K15="$R/tools/kbench_cfg.py --reps 4"
run() { tag=$1; shift; "$@" > $O/$tag.log 2>&1 || echo "$tag: rc=$?" >> $O/errors.txt; }
run stats1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 $B1
run stats3 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3 -- python3 $B3
run stats1536 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1536 -- python3 $R/tools/kbench_cfg.py
run statspipe rocprofv3 --kernel-trace --stats --output-format csv -d $O/statspipe -- python3 $R/tools/kbench_pipe.py
run statspipecopy rocprofv3 --kernel-trace --stats --output-format csv -d $O/statspipecopy -- python3 $R/tools/kbench_pipe.py --copy
run pmc_a rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
  --kernel-trace --output-format csv -d $O/pmc_a -- python3 $KB
run pmc_b rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR \
  --kernel-trace --output-format csv -d $O/pmc_b -- python3 $KB
run fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $KB
run write rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $KB
run pmc1536_a rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
  --kernel-trace --output-format csv -d $O/pmc1536_a -- python3 $K15
run pmc1536_b rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR \
  --kernel-trace --output-format csv -d $O/pmc1536_b -- python3 $K15
run fetch1536 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch1536 -- python3 $K15
run write1536 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write1536 -- python3 $K15
# shader clock under the kernel: diagnostic build with in-kernel stamps (never the product library)
if [ -f $R/dsp-speech-recognition_amd/lib/variants/stamps.so ]; then
  DSP_FRONTEND_LIB=$R/dsp-speech-recognition_amd/lib/variants/stamps.so python3 $R/tools/kbench.py > $O/stamps.log 2>&1
fi
# keep only the small summary files (the raw traces are large)
find $O -name "*_kernel_trace.csv" -size +2M -delete
find $O -name "*.db" -delete
python3 $R/tools/profile_r4_summary.py $O > $O/summary.txt 2>&1
head -150 $O/summary.txt
du -sh $O
