#!/bin/bash
# Round-4 evidence run (on the GPU box through gpurun).  Everything lands under gpurun_out/r04/; tools/refresh_profiles_r04.py copies the
# summaries into profiles/ (tag r04) and rebuilds profiles/pmc_traffic.json.  Sections (pass names to run a subset):
#   legs       `bench.py --workload c1..c4` (through CrossDomainTrainer.fit) + c3 with the literal dense Adam + c4 with the full last layer
#   e2e        `bench.py --only-e2e`: fit() over SOURCE / TARGET / OVERLAP epochs + evaluate at the headline table sizes
#   c5         the default command `python bench.py` (headline + every leg)
#   trace_c5   rocprofv3 --kernel-trace --stats of `bench.py --headline-only`
#   trace_cfg  rocprofv3 --kernel-trace --stats of `--workload c1|c2|c3` (graph replays) and of the CoNet full-sort leg
#   pmc        --pmc FETCH_SIZE / WRITE_SIZE (own passes, --kernel-trace only) over the headline; MFMA-busy over the CoNet full-sort kernel
#   mfma       --pmc SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE over tools/mb_conet.py and tools/mb_mapstep.py
#   mb         micro-benchmarks: cache-resident gather bandwidth, graph-launch gap
# (The per-block stamps of conet_fb_kernel need a -DCDR_CONET_PROF build of cdr_conet.hip linked as another .so and CDR_LIB_PATH / CDR_CONET_PROF=1:
#  profiles/r04_mb_conet_waves.txt names the commands.)
set -u
ulimit -c 0          # (a faulting kernel must not fill the box's disk with a GPU core dump: everything behind it would fail)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
W="${*:-legs e2e c5 trace_c5 trace_cfg pmc mfma mb}"
has() { [[ " $W " == *" $1 "* ]]; }
if has legs; then
  for w in c1 c2 c3 c4; do python bench.py --workload $w --steps 200 --warmup 20 > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"; done
  python bench.py --workload c3 --steps 200 --warmup 20 --dense-adam --no-cpu-baseline --no-fullsort > $O/bench_c3_dense_adam.json 2>/dev/null; echo "c3 dense rc=$?"
  python bench.py --workload c4 --steps 200 --warmup 20 --full-last-layer --no-cpu-baseline > $O/bench_c4_full_last_layer.json 2>/dev/null; echo "c4 full rc=$?"
fi
if has e2e; then python bench.py --only-e2e > $O/bench_e2e.json 2> $O/bench_e2e.err; echo "e2e rc=$?"; fi
if has c5; then python bench.py > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"; fi
if has mb; then
  python tools/mb_cache_gather.py > $O/mb_cache_gather.txt 2>&1; echo "mb_cache_gather rc=$?"
  python tools/mb_graph_gap.py > $O/graph_gap.txt 2>&1; echo "graph_gap rc=$?"
  hipcc --offload-arch=gfx950 -O3 tools/mb_cache_bw.hip -o /tmp/mb_cache_bw && /tmp/mb_cache_bw > $O/mb_cache_bw.txt 2>&1; echo "mb_cache_bw rc=$?"
fi
cd /tmp && export TMPDIR=/tmp
if has trace_c5; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c5_headline -o trace -- python $R/bench.py --headline-only > $O/bench_c5_headline_under_rocprof.json 2> $O/trace_c5_headline.err; echo "trace c5 headline rc=$?"
fi
if has trace_cfg; then
  for Wl in c1 c2 c3 c4; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$Wl -o trace -- python $R/bench.py --workload $Wl --no-cpu-baseline --no-fullsort --steps 200 --warmup 20 > $O/bench_${Wl}_under_rocprof.json 2> $O/trace_$Wl.err; echo "trace $Wl rc=$?"
  done
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fullsort_conet -o trace -- python $R/bench.py --workload c3 --no-cpu-baseline --steps 20 --warmup 2 > $O/bench_fullsort_conet_under_rocprof.json 2> $O/trace_fullsort_conet.err; echo "trace fullsort conet rc=$?"
fi
if has pmc; then
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -o pmc -- python $R/bench.py --no-cpu-baseline --no-fullsort --no-config-legs --no-e2e --single-stream --steps 3 --warmup 1 > $O/bench_pmc_$C.json 2> $O/pmc_$C.err; echo "pmc $C rc=$?"
  done
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_conet_fullsort -o pmc -- python $R/bench.py --workload c3 --no-cpu-baseline --steps 20 --warmup 2 > $O/bench_c3_under_pmc.json 2> $O/pmc_mfma_conet_fullsort.err; echo "pmc mfma conet fullsort rc=$?"
fi
if has mfma; then          # matrix-pipe counters of the CoNet training kernels and the mapping kernels (own passes; refresh files r04_pmc_mfma_conet_map.json)
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_conet -o pmc -- python $R/tools/mb_conet.py > $O/mb_conet_under_pmc.txt 2> $O/pmc_mfma_conet.err; echo "pmc conet rc=$?"
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_mapstep -o pmc -- python $R/tools/mb_mapstep.py > $O/mb_mapstep_under_pmc.txt 2> $O/pmc_mfma_mapstep.err; echo "pmc mapstep rc=$?"
fi
find $O -name "*kernel_trace.csv" -size +6M -delete
find $O -name "*counter_collection.csv" -size +24M -delete
ls $O | head -80
