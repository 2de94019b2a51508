#!/bin/bash
# Collect the SQ counter sets for tools/kbench.py on the GPU box (run via gpurun from the repo root).
# usage: bash tools/pmc_run.sh [tag] [bench script + args, default: tools/kbench.py --reps 10 --rounds 1]
set -e
TAG=${1:-pmc}
shift || true
if [ $# -gt 0 ]; then BENCH="$*"; else BENCH="tools/kbench.py --reps 10 --rounds 1"; fi
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${TAG}_a $R/gpurun_out/${TAG}_b
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
  --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_a -- python3 $R/$BENCH > $R/gpurun_out/${TAG}_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL \
  --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_b -- python3 $R/$BENCH > $R/gpurun_out/${TAG}_b.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_a $R/gpurun_out/${TAG}_b
