#!/bin/bash
# Round-3 evidence run (on the GPU box through gpurun).  Everything lands under gpurun_out/r03/; tools/refresh_profiles_r03.py copies
# the summaries into profiles/ (tag r03) and rebuilds profiles/pmc_traffic.json.
#   1. the default command `python bench.py` (headline + every leg + the compact C1-C4 legs)                  -> bench_c5.json
#   2. rocprofv3 --kernel-trace --stats of THE SAME default command                                          -> trace_c5/*kernel_stats.csv
#      ... and of `python bench.py --headline-only` (the headline kernels on the headline batch alone)       -> trace_c5_headline/
#   3. --pmc FETCH_SIZE / WRITE_SIZE, each in its own pass (--kernel-trace only), over the headline + OVERLAP legs (--steps 3)
#   4. --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over the U = 1,024 scoring kernel (tools/mb_fullsort.py)
#   5. rocprofv3 --kernel-trace --stats of `--workload c1|c2|c3` (graph replays) and `--workload c4` (eager: a traced hipGraph replay of it once hung the profiler)
#   6. micro-benchmarks: the fused step per kernel (uniform / Zipf), both domains on one vs two streams, OVERLAP step, models5
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python bench.py > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-fullsort --no-config-legs > $O/bench_c5_200steps.json 2> /dev/null; echo "c5-200 rc=$?"
python tools/mb_step.py > $O/mb_step.txt 2>&1; python tools/mb_step.py 50000001 20000001 1048576 128 zipf >> $O/mb_step.txt 2>&1; echo "mb_step rc=$?"
CDR_FUSE_SINGLES=0 python tools/mb_step.py >> $O/mb_step.txt 2>&1
python tools/mb_step2.py > $O/mb_step2.txt 2>&1; echo "mb_step2 rc=$?"
python tools/mb_kmajor.py > $O/mb_kmajor.txt 2>&1; echo "mb_kmajor rc=$?"
python tools/mb_mapstep.py > $O/mb_mapstep.txt 2>&1; echo "mb_mapstep rc=$?"
python tools/mb_models5.py > $O/mb_models5.txt 2>&1; echo "mb_models5 rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c5 -o trace -- python $R/bench.py --no-cpu-baseline > $O/bench_c5_under_rocprof.json 2> $O/trace_c5.err; echo "trace c5 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c5_headline -o trace -- python $R/bench.py --headline-only > $O/bench_c5_headline_under_rocprof.json 2> $O/trace_c5_headline.err; echo "trace c5 headline rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3 -o trace -- python $R/bench.py --workload c3 --no-cpu-baseline --steps 100 --warmup 10 > $O/bench_c3_under_rocprof.json 2> $O/trace_c3.err; echo "trace c3 rc=$?"
for W in c1 c2; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$W -o trace -- python $R/bench.py --workload $W --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_${W}_under_rocprof.json 2> $O/trace_$W.err; echo "trace $W rc=$?"
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4 -o trace -- python $R/bench.py --workload c4 --no-cpu-baseline --no-graph --steps 50 --warmup 5 > $O/bench_c4_under_rocprof.json 2> $O/trace_c4.err; echo "trace c4 rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -o pmc -- python $R/bench.py --no-cpu-baseline --no-fullsort --no-config-legs --steps 3 --warmup 1 > $O/bench_pmc_$C.json 2> $O/pmc_$C.err; echo "pmc $C rc=$?"
done
MB_U=1024 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o pmc -- python $R/tools/mb_fullsort.py 128 > $O/mb_fullsort_under_pmc.txt 2> $O/pmc_mfma.err; echo "pmc mfma rc=$?"
for T in conet mapstep; do
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_$T -o pmc -- python $R/tools/mb_$T.py > $O/mb_${T}_under_pmc.txt 2> $O/pmc_mfma_$T.err; echo "pmc mfma $T rc=$?"
done   # (tools/refresh_profiles_r03.py -> profiles/r03_pmc_mfma_conet_map.json: busy / ((active / 8 XCDs) x 1,024 SIMDs) per kernel)
find $O -name "*kernel_trace.csv" -size +6M -delete
find $O -name "*counter_collection.csv" -size +24M -delete
ls -la $O | head -60
