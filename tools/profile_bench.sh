#!/bin/bash
# Profiles bench.py on the GPU box: one rocprofv3 --kernel-trace --stats run and four separate
# --pmc passes (counters never share a run with other trace domains), then the per-kernel summary.
#   tools/profile_bench.sh TAG [bench.py arguments...]        (from the repository root)
# Leaves gpurun_out/prof_TAG/{stats,p1,p2,p3,p4}/ and gpurun_out/prof_TAG/pmc_summary.json.
set -e -o pipefail
TAG=$1; shift
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1
echo "stats pass done"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p1" -o p1 -- python3 "$ROOT/bench.py" $ARGS > "$OUT/p1.log" 2>&1
echo "pmc pass 1 done"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/p2" -o p2 -- python3 "$ROOT/bench.py" $ARGS > "$OUT/p2.log" 2>&1
echo "pmc pass 2 done"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/p3" -o p3 -- python3 "$ROOT/bench.py" $ARGS > "$OUT/p3.log" 2>&1
echo "pmc pass 3 done"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d "$OUT/p4" -o p4 -- python3 "$ROOT/bench.py" $ARGS > "$OUT/p4.log" 2>&1
echo "pmc pass 4 done"
cd "$ROOT"
# bench.py's train workload runs its timed batches twice (headline pass, then the pass with per-kernel
# events): 3 + 10 + 10 steps.  Other workloads: SUMMARY_STEPS=<steps the command runs> (rows per launch position)
python3 tools/pmc_summary.py "$OUT/pmc_summary.json" steps=${SUMMARY_STEPS:-23} mfma="$OUT/p1/p1" fetch="$OUT/p2/p2" write="$OUT/p3/p3" sizes="$OUT/p4/p4" > "$OUT/pmc_summary.txt"
tail -5 "$OUT/stats.log"
