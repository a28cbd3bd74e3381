#!/bin/bash
# A/B of the deferred Adam's replay claiming its rows through last[] in one launch with the id sort's counting pass (CDR_LZ_CLAIM=0: sort, then replay).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
{
python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer_graph.py tests/test_abi.py -q -m gpu -x -k "conet or c3 or abi or deferred" 2>&1 | grep -E "passed|failed|error" | tail -3
for rep in 1 2 3; do for v in 1 0; do
  echo "== CDR_LZ_CLAIM=$v rep $rep: ms_per_step, final loss, abs_total"
  CDR_LZ_CLAIM=$v python bench.py --workload c3 --steps 400 --warmup 40 --no-cpu-baseline --no-fullsort --detail-file $O/lzclaim_$v.json > /dev/null 2>&1; python -c "import json; d=json.load(open('$O/lzclaim_$v.json')); d=d.get('headline', d); print(d['ms_per_step'], d.get('final_loss'), (d.get('state_checksum') or {}).get('abs_total'))"
done; done
} > $O/ab_lz_claim.txt 2>&1
cat $O/ab_lz_claim.txt
