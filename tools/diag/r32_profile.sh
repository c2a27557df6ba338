#!/bin/bash
# Diagnostic (GPU box): where the reference-exact 32x32 model's step goes - isolated HIP-event table, single-stream step, rocprofv3 kernel stats.
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r32
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --size 32 --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
python3 $B > $OUT/bench.json 2> /dev/null
python3 $B --set use_side_stream=0 > $OUT/bench_single_stream.json 2> /dev/null
python3 $B --kernels --set use_side_stream=0 > /dev/null 2> $OUT/per_kernel_isolated.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $B > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/trace -name "*kernel_trace.csv" -size +30M -delete
echo done
