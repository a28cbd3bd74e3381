cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c; mkdir -p $O
FS="--force-shard --single-layout --no-fullsort --no-cpu-baseline --no-config-legs --no-e2e --no-ingest --steps 40 --warmup 5 --shard row"
run() { name=$1; shift; env "$@" timeout 600 python bench.py $FS --detail-file $O/$name.json 2> $O/$name.err | tail -1 > $O/${name}_line.json
python -c "
import json;d=json.load(open('$O/${name}_line.json'));print('$name', d['ms_per_step'],d['value'])"; }
run base A=1
run nt CDR_LIB_PATH=$GRAFT_REPO_ROOT/tools/r06/ab/libcdrhip_nt.so
run prio CDR_SIDE_PRIO=-1
run nt_prio CDR_SIDE_PRIO=-1 CDR_LIB_PATH=$GRAFT_REPO_ROOT/tools/r06/ab/libcdrhip_nt.so
run base2 A=1
run nt2 CDR_LIB_PATH=$GRAFT_REPO_ROOT/tools/r06/ab/libcdrhip_nt.so
# the unsharded headline with the nt build
for lib in "" "$GRAFT_REPO_ROOT/tools/r06/ab/libcdrhip_nt.so"; do
CDR_LIB_PATH=$lib timeout 600 python bench.py --no-fullsort --no-cpu-baseline --no-config-legs --no-e2e --no-ingest --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('headline lib=[$lib]', d['ms_per_step'],d['value'], d['roofline'])"
done
