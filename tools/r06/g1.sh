set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
O=gpurun_out/r06a
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sharded_step_world1 or sharded_native_ranks" 2>&1 | tail -15 > $O/tests.txt
cat $O/tests.txt
FS="--force-shard --single-layout --no-fullsort --no-cpu-baseline --no-config-legs --no-e2e --no-ingest --steps 10 --warmup 3"
timeout 600 python bench.py $FS --shard row --detail-file $O/force_shard_row.json 2> $O/force_shard_row.err | tail -1 > $O/force_shard_row_line.json; echo rc=$?
tail -3 $O/force_shard_row.err
python -c "
import json;d=json.load(open('$O/force_shard_row_line.json'));print(d['ms_per_step'],d['value'])"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_row -o trace -- python $GRAFT_REPO_ROOT/bench.py $FS --shard row --no-map > /dev/null 2> $GRAFT_REPO_ROOT/$O/trace_row.err; echo trace rc=$?
cd $GRAFT_REPO_ROOT
find $O/trace_row -name "*kernel_stats.csv" | head; find $O/trace_row -name "*kernel_trace.csv" -delete; find $O/trace_row -name "*.db" -delete
ls -la $O/trace_row/* | head
