cd $GRAFT_REPO_ROOT
O=gpurun_out/r06h; mkdir -p $O
tools/r06/ab/mb_atomics | tee $O/mb_atomics.txt
FS="--force-shard --single-layout --no-fullsort --no-cpu-baseline --no-config-legs --no-e2e --no-ingest --steps 20 --warmup 5 --shard row"
run() { name=$1; shift; env "$@" timeout 600 python bench.py $FS --detail-file $O/$name.json 2> $O/$name.err | tail -1 > $O/${name}_line.json
python -c "
import json;d=json.load(open('$O/${name}_line.json'));print('$name', d['ms_per_step'],d['value'])"; }
run row A=1
FS="$FS --comm cabi"
run row_cabi A=1
python tools/mb_dimshard.py > $O/mb_dimshard.json 2> $O/mb_dimshard.err; echo "mb_dimshard rc=$?"
timeout 900 python -m pytest tests/test_gpu_trainer_graph.py -x -q -m gpu -k "conet" 2>&1 | tail -4
timeout 600 python bench.py --steps 20 --warmup 5 --no-config-legs --no-e2e --no-ingest --no-fullsort --detail-file $O/head.json 2>/dev/null | tail -1 > $O/head_line.json; python -c "
import json;d=json.load(open('$O/head_line.json'));print(d['ms_per_step'], d['cpu_baseline'], d.get('vs_cpu'))"
