# A/B harness of round 6 (run on the GPU box through gpurun).  The variant library is built out of tree from a copy of csrc/ with the macro named
# in DESIGN.md / csrc (e.g. -DCDR_NO_STREAM_NT, -DCDR_NO_MAP_NT) and LIB=tools/r06/ab/<name>.so; CDR_LIB_PATH selects it per run.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d; mkdir -p $O
NT=$GRAFT_REPO_ROOT/tools/r06/ab/libcdrhip_nt.so
FS="--force-shard --single-layout --no-fullsort --no-cpu-baseline --no-config-legs --no-e2e --no-ingest --steps 40 --warmup 5 --shard row"
HL="--no-fullsort --no-cpu-baseline --no-config-legs --no-e2e --no-ingest --steps 40 --warmup 5"
for i in 1 2 3 4; do
for lib in "" "$NT"; do
CDR_LIB_PATH=$lib timeout 600 python bench.py $FS 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('row  nt=%d' % bool('$lib'), round(d['ms_per_step'],3))"
CDR_LIB_PATH=$lib timeout 600 python bench.py $HL 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('head nt=%d' % bool('$lib'), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],3))"
done; done 2>&1 | tee $O/ab_nt.txt
