ulimit -c 0
cd /root/repo
O=gpurun_out/fork; mkdir -p $O
for f in 1 0; do
  CDR_STEP_FORK=$f python bench.py --no-cpu-baseline --no-fullsort --no-config-legs --no-e2e --steps 30 --warmup 5 > $O/two_$f.json 2> $O/two_$f.err
  CDR_STEP_FORK=$f python bench.py --no-cpu-baseline --no-fullsort --no-config-legs --no-e2e --single-stream --steps 30 --warmup 5 > $O/one_$f.json 2> $O/one_$f.err
done
python - <<'PY'
import json
for n in ('two_1','two_0','one_1','one_0'):
    try:
        d=json.loads(open('gpurun_out/fork/%s.json'%n).read().strip().splitlines()[-1])
        g=d.get('synthetic_grid',{})
        print(n, d['value']/1e6, d['ms_per_step'], d.get('single_stream'), {k:(v.get('ms_per_domain_step'),v.get('hipgraph_ms_per_domain_step')) for k,v in g.items()} if isinstance(g,dict) else None)
    except Exception as e: print(n,'ERR',e)
PY
cd /tmp; export TMPDIR=/tmp
NU=8000001 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /root/repo/$O/pmc_map -o pmc -- python /root/repo/tools/mb_mapstep.py > /root/repo/$O/mb_mapstep_pmc.txt 2>&1; echo pmc rc=$?
cd /root/repo; python -m pytest tests -m gpu -q -x -k "step or fused or rowwise or trainer" 2>&1 | tail -3
