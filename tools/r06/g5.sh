cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; mkdir -p $O
python tools/mb_dimshard.py > $O/mb_dimshard.json 2> $O/mb_dimshard.err; echo "mb_dimshard rc=$?"
CDR_LIB_PATH=$GRAFT_REPO_ROOT/tools/r06/ab/libcdrhip_plain.so python tools/mb_dimshard.py > $O/mb_dimshard_plain.json 2> /dev/null; echo "mb_dimshard plain rc=$?"
FS="--force-shard --single-layout --no-fullsort --no-cpu-baseline --no-config-legs --no-e2e --no-ingest --steps 20 --warmup 5 --shard row"
run() { name=$1; shift; env "$@" timeout 600 python bench.py $FS --detail-file $O/$name.json 2> $O/$name.err | tail -1 > $O/${name}_line.json
python -c "
import json;d=json.load(open('$O/${name}_line.json'));print('$name', d['ms_per_step'],d['value'])"; }
run row A=1
run row_rccl CDR_A2A_SELF_VIA_RCCL=1
FS="$FS --comm cabi"
run row_cabi A=1
FS="--force-shard --single-layout --no-fullsort --no-cpu-baseline --no-config-legs --no-e2e --no-ingest --steps 20 --warmup 5 --shard dim"
run dim A=1
