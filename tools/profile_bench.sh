#!/bin/bash
# rocprofv3 passes over the default bench command (run on the GPU box through gpurun):
#   1. --kernel-trace --stats          -> per-kernel durations (profiles/<tag>_kernel_stats.csv)
#   2. --pmc FETCH_SIZE  (own pass)    -> HBM read traffic per dispatch
#   3. --pmc WRITE_SIZE  (own pass)    -> HBM write traffic per dispatch
# usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python "$REPO/bench.py" --no-cpu-baseline "$@" > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.err"
echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o pmc -- python "$REPO/bench.py" --no-cpu-baseline --no-fullsort --steps 3 --warmup 1 > "$OUT/bench_pmc_$C.json" 2> "$OUT/bench_pmc_$C.err"
  echo "pmc $C rc=$?"
done
find "$OUT" -type f | head -40
F=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && head -12 "$F"
python "$REPO/tools/summarize_pmc.py" "$OUT" > "$OUT/pmc_summary.json" 2> "$OUT/pmc_summary.err"; cat "$OUT/pmc_summary.json"; tail -3 "$OUT/pmc_summary.err"
# keep the big raw traces out of the merge-back (64 MiB cap): drop per-dispatch kernel traces, keep stats + counters summary
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
find "$OUT" -name "*counter_collection.csv" -size +8M -delete
