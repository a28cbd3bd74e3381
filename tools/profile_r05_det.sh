# Round-5 late evidence: the driver's command on the final build, and rocprofv3 --kernel-trace --stats of the C1 / C2 legs under
# CDR_DETERMINISTIC=1 (cdr_ordered_bwd in the replayed step).  Output: gpurun_out/r05g/
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05g; mkdir -p $O
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 --detail-file $O/bench_c5.json 2> $O/bench_c5.err | tail -n 1 > $O/bench_c5_line.json; echo "c5 rc=$? stderr lines: $(grep -vc amdgpu.ids $O/bench_c5.err)"
cd /tmp && export TMPDIR=/tmp
for Wl in c1 c2; do
  CDR_DETERMINISTIC=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_${Wl}_det -o trace -- python $R/bench.py --workload $Wl --no-cpu-baseline --no-fullsort --steps 200 --warmup 20 --detail-file $O/bench_${Wl}_det_under_rocprof.json > /dev/null 2> $O/trace_${Wl}_det.err; echo "trace $Wl det rc=$?"
  cp $(find $O/trace_${Wl}_det -name "*kernel_stats.csv" | head -1) $O/bench_${Wl}_det_kernel_stats.csv
done
cd $R
head -c 600 $O/bench_c5_line.json; echo
for Wl in c1 c2; do grep -i "ordered\|fwd\|adam_multi\|batch_produce" $O/bench_${Wl}_det_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160; done
