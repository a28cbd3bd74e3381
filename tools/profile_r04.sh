#!/bin/bash
# Round-4 evidence run (on the GPU box through gpurun).  Everything lands under gpurun_out/r04/; tools/refresh_profiles_r04.py copies the
# summaries into profiles/ (tag r04).  Pass a list of section names to run only those: legs e2e c5 trace_c5 trace_cfg pmc mb
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
W="${*:-legs e2e c5 trace_c5 trace_cfg}"
has() { [[ " $W " == *" $1 "* ]]; }
if has legs; then
  for w in c1 c2 c3 c4; do python bench.py --workload $w --steps 200 --warmup 20 > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"; done
  python bench.py --workload c3 --steps 200 --warmup 20 --dense-adam --no-cpu-baseline --no-fullsort > $O/bench_c3_dense_adam.json 2>/dev/null; echo "c3 dense rc=$?"
  python bench.py --workload c4 --steps 200 --warmup 20 --full-last-layer --no-cpu-baseline > $O/bench_c4_full_last_layer.json 2>/dev/null; echo "c4 full rc=$?"
fi
if has e2e; then python bench.py --only-e2e > $O/bench_e2e.json 2> $O/bench_e2e.err; echo "e2e rc=$?"; fi
if has c5; then python bench.py > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"; fi
cd /tmp && export TMPDIR=/tmp
if has trace_c5; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c5_headline -o trace -- python $R/bench.py --headline-only > $O/bench_c5_headline_under_rocprof.json 2> $O/trace_c5_headline.err; echo "trace c5 headline rc=$?"
fi
if has trace_cfg; then
  for Wl in c1 c2 c3; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$Wl -o trace -- python $R/bench.py --workload $Wl --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_${Wl}_under_rocprof.json 2> $O/trace_$Wl.err; echo "trace $Wl rc=$?"
  done
fi
find $O -name "*kernel_trace.csv" -size +6M -delete
find $O -name "*counter_collection.csv" -size +24M -delete
ls $O | head -80
