cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06q
P=$GRAFT_REPO_ROOT/tools/r06/ab/libcdrhip_nomapnt.so
for i in 1 2 3; do for lib in "" "$P"; do echo "== nt=$([ -z "$lib" ] && echo 1 || echo 0)"; CDR_LIB_PATH=$lib python tools/mb_mapstep.py 2>/dev/null | grep -E "65536" | cut -c1-170; done; done | tee gpurun_out/r06q/ab_map_nt.txt
