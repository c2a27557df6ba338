#!/bin/bash
# Regenerate the round's profile artifacts on the GPU box (one gpurun call):
#   gpurun --timeout 1100 -- 'bash tools/make_profiles.sh r02 v3'
# then, back in the container:  bash tools/collect_profiles.sh r02 v3
# Everything lands under gpurun_out/prof_<round>_<ver>/ (merged back by gpurun).
set -e -o pipefail
RND=$1; VER=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${RND}_${VER}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="$ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras"
# 1. the bench line itself (with the cpu baseline and the extra records) + HIP-event tables + launch order
python3 $ROOT/bench.py --steps 30 --warmup 5 --kernels --dump-order $OUT/order.json > $OUT/bench.json 2> $OUT/per_kernel_hip_events.txt
python3 $BENCH --kernels --set use_side_stream=0 > /dev/null 2> $OUT/per_kernel_isolated.txt
echo "bench done"
# 2. rocprofv3 kernel trace + stats of the bench command (no counters)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $BENCH > $OUT/trace.log 2>&1
echo "trace done"
# 3. HBM traffic: separate counter passes, kernel trace only
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $BENCH --steps 6 --warmup 3 > $OUT/pmc_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $BENCH --steps 6 --warmup 3 > $OUT/pmc_write.log 2>&1
echo "write pass done"
# 4. event timeline of one step (streams overlapping)
python3 $ROOT/tools/diag/gpu_timeline.py > $OUT/timeline.txt 2>/dev/null
# keep the merge small: counter CSVs are large, the per-launch table is what is needed
python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/order.json 128 16 256 bf16 --out $OUT/pmc_traffic.json
find $OUT -name "*counter_collection.csv" -size +20M -delete
echo "profiles done"
