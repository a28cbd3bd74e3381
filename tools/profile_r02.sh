#!/bin/bash
# Round-2 evidence run (on the GPU box through gpurun): bench lines of every BASELINE config, rocprofv3 kernel stats of the c5 and
# c3 commands, PMC passes (FETCH_SIZE / WRITE_SIZE in their own runs, --kernel-trace only), the sweeps and micro-benchmarks.
# (rocprofv3 runs sit under `timeout`: a traced hipGraph replay of the C4 step once hung the profiler for the whole call.)
# Everything lands under gpurun_out/r02/; tools/refresh_profiles_r02.py copies the summaries into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02; mkdir -p $O
cd $R
python bench.py > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-fullsort > $O/bench_c5_200steps.json 2> /dev/null; echo "c5-200 rc=$?"
for w in c1 c2 c3 c4; do python bench.py --workload $w --steps 200 --warmup 20 > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"; done
python bench.py --workload c3 --dense-adam --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_c3_dense_adam.json 2> /dev/null; echo "c3-dense rc=$?"
python bench.py --workload c4 --full-last-layer --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_c4_full_last_layer.json 2> /dev/null; echo "c4-full rc=$?"
python tools/sweep_small.py > $O/sweep_small.txt 2>&1; echo "sweep_small rc=$?"
python tools/mb_kmajor.py > $O/mb_kmajor.txt 2>&1; echo "mb_kmajor rc=$?"
python tools/mb_mapstep.py > $O/mb_mapstep.txt 2>&1; echo "mb_mapstep rc=$?"
python tools/mb_conet.py > $O/mb_conet.txt 2>&1; echo "mb_conet rc=$?"
python tools/mb_smallsort.py > $O/mb_smallsort.txt 2>&1; echo "mb_smallsort rc=$?"
python tools/mb_models5.py > $O/mb_models5.txt 2>&1; echo "mb_models5 rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c5 -o trace -- python $R/bench.py --no-cpu-baseline > $O/bench_c5_under_rocprof.json 2> $O/trace_c5.err; echo "trace c5 rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3 -o trace -- python $R/bench.py --workload c3 --no-cpu-baseline --steps 100 --warmup 10 > $O/bench_c3_under_rocprof.json 2> $O/trace_c3.err; echo "trace c3 rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -o pmc -- python $R/bench.py --no-cpu-baseline --no-fullsort --steps 3 --warmup 1 > $O/bench_pmc_$C.json 2> $O/pmc_$C.err; echo "pmc $C rc=$?"
done
find $O -name "*kernel_trace.csv" -size +6M -delete
find $O -name "*counter_collection.csv" -size +24M -delete
ls -la $O | head -50
