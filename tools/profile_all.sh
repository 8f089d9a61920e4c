#!/bin/bash
# Runs tools/profile_bench.sh for every bench mode and copies the summaries that are kept under
# profiles/ (round tag as first argument, e.g. r02):  tools/profile_all.sh r02
set -e -o pipefail
R=$1
ROOT=$PWD
mkdir -p profiles gpurun_out
run() {  # name-in-profiles  tag  [bench args...]
    local name=$1 tag=$2; shift 2
    bash tools/profile_bench.sh "$tag" "$@" > "gpurun_out/prof_$tag.log" 2>&1
    cp "gpurun_out/prof_$tag/stats/stats_kernel_stats.csv" "profiles/${R}_${name}bench_kernel_stats.csv"
    cp "gpurun_out/prof_$tag/pmc_summary.json" "profiles/${R}_${name}pmc_summary.json"
    echo "profiled $tag"
}
# PART=a: the four train modes under the profiler; PART=b: render, bf16, the reference's own batch
# size and the unprofiled bench lines (a gpurun call is limited to 20 minutes); unset: everything
if [ "${PART:-a}" = a ]; then
run "" "${R}_exact"
run "split_" "${R}_split" --math split-bf16
run "fused_" "${R}_fused" --fuse-linear-pair
run "fused_split_" "${R}_fused_split" --fuse-linear-pair --math split-bf16
fi
if [ "${PART:-b}" != b ]; then echo "part a done"; exit 0; fi
SUMMARY_STEPS=0 run "render_" "${R}_render" --workload render
SUMMARY_STEPS=16 run "bf16_cfg5_" "${R}_bf16" --workload bf16 --micro-batches 4
# the reference's own batch (16 rays x 128 samples, train.c:16-17): kernel times only
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv \
    -d "$ROOT/gpurun_out/prof_${R}_ref16" -o stats -- python3 "$ROOT/bench.py" --rays 16 --steps 2000 --warmup 200 \
    --no-cpu-baseline --no-secondary > "$ROOT/gpurun_out/prof_${R}_ref16.log" 2>&1 )
cp "gpurun_out/prof_${R}_ref16/stats_kernel_stats.csv" "profiles/${R}_ref16_kernel_stats.csv"
grep "^{\"metric" "gpurun_out/prof_${R}_ref16.log" > "profiles/${R}_ref16_bench_line_under_profiler.json"
python3 bench.py --rays 16 --steps 2000 --warmup 200 --no-cpu-baseline --no-secondary > "profiles/${R}_ref16_bench_line.json" 2>> gpurun_out/bench_line.err
# the bench lines themselves, unprofiled
python3 bench.py > "profiles/${R}_bench_line.json" 2> gpurun_out/bench_line.err
python3 bench.py --no-secondary --no-cpu-baseline --math split-bf16 > "profiles/${R}_split_bench_line.json" 2>> gpurun_out/bench_line.err
python3 bench.py --no-secondary --no-cpu-baseline --fuse-linear-pair > "profiles/${R}_fused_bench_line.json" 2>> gpurun_out/bench_line.err
python3 bench.py --no-secondary --no-cpu-baseline --fuse-linear-pair --math split-bf16 > "profiles/${R}_fused_split_bench_line.json" 2>> gpurun_out/bench_line.err
python3 bench.py --workload render --fuse-linear-pair --math split-bf16 > "profiles/${R}_bench_line_render_fused_split.json" 2>> gpurun_out/bench_line.err
python3 tools/composite_bench.py > "profiles/${R}_composite_bench.txt" 2>> gpurun_out/bench_line.err
python3 tools/bf16_bench.py > "profiles/${R}_bf16_products.txt" 2>> gpurun_out/bench_line.err
mkdir -p gpurun_out/profiles_$R && cp profiles/${R}_* gpurun_out/profiles_$R/
echo "all done"
