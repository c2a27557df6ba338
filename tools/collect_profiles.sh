#!/bin/bash
# Copy what tools/make_profiles.sh produced (gpurun_out/prof_<round>_<ver>/) into profiles/ under per-round names and
# write the one-page summary.   bash tools/collect_profiles.sh r02 v3
set -e
RND=$1; VER=$2
SRC=gpurun_out/prof_${RND}_${VER}
P=profiles
cp $SRC/bench.json $P/${RND}_bench_${VER}.json
cp $SRC/per_kernel_hip_events.txt $P/${RND}_bench_${VER}_per_kernel_hip_events.txt
cp $SRC/per_kernel_isolated.txt $P/${RND}_bench_${VER}_per_kernel_isolated.txt
cp $SRC/timeline.txt $P/${RND}_timeline_${VER}.txt
cp $(find $SRC/trace -name "*kernel_stats.csv" | head -1) $P/${RND}_bench_h128_b256_bf16_kernel_stats_${VER}.csv
cp $SRC/pmc_traffic.json $P/pmc_traffic.json
python3 tools/trace_timeline.py $SRC/trace > $P/${RND}_rocprof_step_${VER}.txt
python3 tools/profile_summary.py ${RND}_${VER} $P/${RND}_bench_h128_b256_bf16_kernel_stats_${VER}.csv $P/${RND}_bench_${VER}.json $P/${RND}_bench_${VER}_per_kernel_hip_events.txt 128 16 256 bf16
ls -la $P | grep ${RND}
