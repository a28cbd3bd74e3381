# A/B harness of round 6 (run on the GPU box through gpurun).  The variant library is built out of tree from a copy of csrc/ with the macro named
# in DESIGN.md / csrc (e.g. -DCDR_NO_STREAM_NT, -DCDR_NO_MAP_NT) and LIB=tools/r06/ab/<name>.so; CDR_LIB_PATH selects it per run.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06q
P=$GRAFT_REPO_ROOT/tools/r06/ab/libcdrhip_nomapnt.so
for i in 1 2 3; do for lib in "" "$P"; do echo "== nt=$([ -z "$lib" ] && echo 1 || echo 0)"; CDR_LIB_PATH=$lib python tools/mb_mapstep.py 2>/dev/null | grep -E "65536" | cut -c1-170; done; done | tee gpurun_out/r06q/ab_map_nt.txt
