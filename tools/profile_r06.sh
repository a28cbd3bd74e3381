#!/bin/bash
# Round-6 evidence run (on the GPU box through gpurun).  Everything lands under gpurun_out/r06/; tools/refresh_profiles_r06.py copies the
# summaries into profiles/ (tag r06) and rebuilds profiles/pmc_traffic.json.  Sections (pass names to run a subset):
#   c5         the driver's command `python bench.py --gpus 1 --steps 20 --warmup 5`: last stdout line (the bounded contract line) + bench_detail.json
#   legs       `bench.py --workload c1..c4` (through CrossDomainTrainer.fit) + c3 with the literal dense Adam + c4 with the full last layer
#   e2e        `bench.py --only-e2e`
#   ingest     `bench.py --only-ingest` + rocprofv3 --kernel-trace --stats of it (the device overlap remap, csrc/cdr_remap_dev.hip)
#   shard      `bench.py --force-shard --shard row|dim` at N = 1 (+ CDR_A2A_SELF_VIA_RCCL=1 for the row layout), rocprofv3 --stats of the row
#              layout's step, tools/mb_dimshard.py (kernel side of the dimension layout at the N = 1 / 2 / 4 / 8 shapes)
#   trace_c5   rocprofv3 --kernel-trace --stats of `bench.py --headline-only`
#   trace_cfg  rocprofv3 --kernel-trace --stats of `--workload c1|c2|c3|c4` (graph replays) and of the CoNet full-sort leg
#   pmc        --pmc FETCH_SIZE / WRITE_SIZE (own passes, --kernel-trace only) over the headline
#   mfma       --pmc SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE over tools/mb_conet.py, tools/mb_mapstep.py and bench.py --workload c3
#   score      --pmc SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE over the U = 1,024 scoring kernel (tools/mb_fullsort.py D = 128 and 64) and the fused
#              mask + top-10 (tools/mb_fullsort_topk.py), + their HIP-event timings without the profiler
#   idpath     tools/mb_idpath.py (medium batches: sorted ids against the count path)
#   tests      python -m pytest tests -m gpu -q
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06; mkdir -p $O
cd $R
W="${*:-c5 legs e2e ingest shard trace_c5 trace_cfg pmc mfma score idpath tests}"
has() { [[ " $W " == *" $1 "* ]]; }
line() { tail -n 1; }
if has c5; then python bench.py --gpus 1 --steps 20 --warmup 5 --detail-file $O/bench_c5.json 2> $O/bench_c5.err | line > $O/bench_c5_line.json; echo "c5 rc=$? stderr lines: $(grep -vc amdgpu.ids $O/bench_c5.err)"; fi
if has legs; then
  for w in c1 c2 c3 c4; do python bench.py --workload $w --steps 200 --warmup 20 --detail-file $O/bench_$w.json 2> $O/bench_$w.err | line > $O/bench_${w}_line.json; echo "$w rc=$?"; done
  python bench.py --workload c3 --steps 200 --warmup 20 --dense-adam --no-cpu-baseline --no-fullsort --detail-file $O/bench_c3_dense_adam.json > /dev/null 2>&1; echo "c3 dense rc=$?"
  python bench.py --workload c4 --steps 200 --warmup 20 --full-last-layer --no-cpu-baseline --detail-file $O/bench_c4_full_last_layer.json > /dev/null 2>&1; echo "c4 full rc=$?"
fi
if has e2e; then python bench.py --only-e2e > $O/bench_e2e.json 2> $O/bench_e2e.err; echo "e2e rc=$?"; fi
if has ingest; then python bench.py --only-ingest > $O/bench_ingest.json 2> $O/bench_ingest.err; echo "ingest rc=$?"; fi
if has shard_cabi; then
  FS="--force-shard --single-layout --no-fullsort --no-cpu-baseline --no-config-legs --no-e2e --no-ingest --steps 10 --warmup 3"
  python bench.py $FS --shard row --comm cabi --detail-file $O/force_shard_row_cabi.json 2> $O/force_shard_row_cabi.err | line > $O/force_shard_row_cabi_line.json; echo "force-shard row cabi rc=$?"
fi
if has shard; then
  FS="--force-shard --single-layout --no-fullsort --no-cpu-baseline --no-config-legs --no-e2e --no-ingest --steps 20 --warmup 5"
  python bench.py $FS --shard row --detail-file $O/force_shard_row.json 2> $O/force_shard_row.err | line > $O/force_shard_row_line.json; echo "force-shard row rc=$?"
  python bench.py $FS --shard row --no-prefetch --detail-file $O/force_shard_row_no_prefetch.json 2> /dev/null | line > $O/force_shard_row_no_prefetch_line.json; echo "force-shard row no-prefetch rc=$?"
  python bench.py $FS --shard row --no-direct --detail-file $O/force_shard_row_staged_r5.json 2> /dev/null | line > $O/force_shard_row_staged_r5_line.json; echo "force-shard row staged (round 5 form) rc=$?"
  python bench.py $FS --shard dim --detail-file $O/force_shard_dim.json 2> $O/force_shard_dim.err | line > $O/force_shard_dim_line.json; echo "force-shard dim rc=$?"
  CDR_A2A_SELF_VIA_RCCL=1 python bench.py $FS --shard row --detail-file $O/force_shard_row_via_rccl.json 2> /dev/null | line > $O/force_shard_row_via_rccl_line.json; echo "force-shard row via rccl rc=$?"
  python bench.py $FS --shard row --comm cabi --detail-file $O/force_shard_row_cabi.json 2> /dev/null | line > $O/force_shard_row_cabi_line.json; echo "force-shard row cabi rc=$?"
  python bench.py $FS --shard row --no-dedup --detail-file $O/force_shard_row_no_dedup.json 2> /dev/null | line > /dev/null; echo "force-shard row no-dedup rc=$?"
  python tools/mb_dimshard.py > $O/mb_dimshard.json 2> $O/mb_dimshard.err; echo "mb_dimshard rc=$?"
fi
if has idpath; then python tools/mb_idpath.py > $O/mb_idpath.json 2> $O/mb_idpath.err; echo "mb_idpath rc=$?"; fi
if has score; then
  for Dd in 128 64; do MB_U=256,1024 python tools/mb_fullsort.py $Dd > $O/mb_fullsort_d$Dd.txt 2>/dev/null; echo "mb_fullsort $Dd rc=$?"; done
  for Uu in 256 1024; do MB_U=$Uu python tools/mb_fullsort_topk.py 128 > $O/mb_fullsort_topk_u$Uu.txt 2>/dev/null; echo "mb_fullsort_topk U=$Uu rc=$?"; done
fi
if has tests; then python -m pytest tests -m gpu -q > $O/gputests.txt 2>&1; echo "gputests rc=$?"; tail -3 $O/gputests.txt; fi
cd /tmp && export TMPDIR=/tmp
if has shard; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_force_shard_row -o trace -- python $R/bench.py $FS --shard row --no-map > /dev/null 2> $O/trace_force_shard_row.err; echo "trace force-shard row rc=$?"
fi
if has ingest; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ingest -o trace -- python $R/bench.py --only-ingest --no-cpu-baseline > $O/bench_ingest_under_rocprof.json 2> $O/trace_ingest.err; echo "trace ingest rc=$?"
fi
if has trace_c5; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c5_headline -o trace -- python $R/bench.py --headline-only --detail-file $O/bench_c5_headline_under_rocprof.json > $O/bench_c5_headline_under_rocprof_line.json 2> $O/trace_c5_headline.err; echo "trace c5 headline rc=$?"
fi
if has trace_cfg; then
  for Wl in c1 c2 c3 c4; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$Wl -o trace -- python $R/bench.py --workload $Wl --no-cpu-baseline --no-fullsort --steps 200 --warmup 20 --detail-file $O/bench_${Wl}_under_rocprof.json > /dev/null 2> $O/trace_$Wl.err; echo "trace $Wl rc=$?"
  done
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fullsort_conet -o trace -- python $R/bench.py --workload c3 --no-cpu-baseline --steps 20 --warmup 2 --detail-file $O/bench_fullsort_conet_under_rocprof.json > /dev/null 2> $O/trace_fullsort_conet.err; echo "trace fullsort conet rc=$?"
fi
if has pmc; then
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -o pmc -- python $R/bench.py --no-cpu-baseline --no-fullsort --no-config-legs --no-e2e --no-ingest --single-stream --steps 3 --warmup 1 --detail-file $O/bench_pmc_$C.json > /dev/null 2> $O/pmc_$C.err; echo "pmc $C rc=$?"
  done
fi
if has score; then
  for Dd in 128 64; do
    MB_U=1024 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_score_d$Dd -o pmc -- python $R/tools/mb_fullsort.py $Dd > $O/mb_fullsort_d${Dd}_under_pmc.txt 2> $O/pmc_score_d$Dd.err; echo "pmc score D=$Dd rc=$?"
  done
  MB_U=1024 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_score_topk -o pmc -- python $R/tools/mb_fullsort_topk.py 128 > $O/mb_fullsort_topk_under_pmc.txt 2> $O/pmc_score_topk.err; echo "pmc score topk rc=$?"
fi
if has mfma; then
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_conet -o pmc -- python $R/tools/mb_conet.py > $O/mb_conet_under_pmc.txt 2> $O/pmc_mfma_conet.err; echo "pmc conet rc=$?"
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_mapstep -o pmc -- python $R/tools/mb_mapstep.py > $O/mb_mapstep_under_pmc.txt 2> $O/pmc_mfma_mapstep.err; echo "pmc mapstep rc=$?"
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_c3 -o pmc -- python $R/bench.py --workload c3 --no-cpu-baseline --steps 20 --warmup 2 --detail-file $O/bench_c3_under_pmc.json > /dev/null 2> $O/pmc_mfma_c3.err; echo "pmc mfma c3 rc=$?"
fi
find $O -name "*kernel_trace.csv" -size +6M -delete
find $O -name "*counter_collection.csv" -size +24M -delete
ls $O | head -100
