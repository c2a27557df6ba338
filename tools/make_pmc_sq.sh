#!/bin/bash
# SQ counters per kernel for the bench workload, side streams off (every launch alone), two counter passes.
#   gpurun --timeout 900 -- 'bash tools/make_pmc_sq.sh v2'   then   python tools/pmc_sq_table.py gpurun_out/pmc_sq_v2 > profiles/r02_pmc_sq_v2.txt
set -e -o pipefail
VER=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_sq_$VER
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VAE_NO_SIDE_STREAM=1
BENCH="$ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras --set use_side_stream=0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/passA -- python3 $BENCH > $OUT/passA.log 2>&1
echo "pass A done"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d $OUT/passB -- python3 $BENCH > $OUT/passB.log 2>&1
echo "pass B done"
find $OUT -name "*kernel_trace.csv" -delete
