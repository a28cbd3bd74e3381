#!/bin/bash
# A/B of the replay's moving window (CDR_LZ_SWEEP = its period in updates; 0: off) -- no row more than ~period updates behind, which bounds the launch's tail.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
{
python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer_graph.py tests/test_abi.py -q -m gpu -x -k "conet or c3 or abi or deferred" 2>&1 | grep -E "passed|failed|error" | tail -3
for rep in 1 2 3; do for v in 128 0 64 256; do
  echo "== CDR_LZ_SWEEP=$v rep $rep: ms_per_step, final loss, abs_total"
  CDR_LZ_SWEEP=$v python bench.py --workload c3 --steps 800 --warmup 400 --no-cpu-baseline --no-fullsort --detail-file $O/lzsweep_$v.json > /dev/null 2>&1; python -c "import json; d=json.load(open('$O/lzsweep_$v.json')); d=d.get('headline', d); print(d['ms_per_step'], d.get('final_loss'), (d.get('state_checksum') or {}).get('abs_total'))"
done; done
} > $O/ab_lz_sweep.txt 2>&1
cat $O/ab_lz_sweep.txt
