set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sharded_step_world1 or sharded_native_ranks" 2>&1 | tail -15 > $O/tests.txt
cat $O/tests.txt
FS="--force-shard --single-layout --no-fullsort --no-cpu-baseline --no-config-legs --no-e2e --no-ingest --steps 10 --warmup 3"
for v in "" "--no-prefetch" "--no-pipeline" "--no-direct"; do
timeout 600 python bench.py $FS --shard row $v --detail-file $O/fsr$v.json 2> $O/fsr$v.err | tail -1 > $O/fsr${v}_line.json; echo rc=$?
python -c "
import json;d=json.load(open('$O/fsr${v}_line.json'));print('$v', d['ms_per_step'],d['value'])"
done
timeout 600 python bench.py $FS --shard row --steps 40 --detail-file $O/fsr40.json 2> /dev/null | tail -1 > $O/fsr40_line.json
python -c "
import json;d=json.load(open('$O/fsr40_line.json'));print('40 steps', d['ms_per_step'],d['value'])"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_row -o trace -- python $GRAFT_REPO_ROOT/bench.py $FS --shard row --no-map > /dev/null 2> $GRAFT_REPO_ROOT/$O/trace_row.err; echo trace rc=$?
cd $GRAFT_REPO_ROOT
find $O/trace_row -name "*kernel_trace.csv" -delete; find $O/trace_row -name "*.db" -delete
