#!/bin/bash
# A/B of CoNet's loss total riding in the backward's weight-gradient launch inside the pipelined captured step (CDR_CONET_DEFER_FINISH=0: the
# forward's own finishing launch).  Output: gpurun_out/r06/ab_defer_finish.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
{
python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer_graph.py tests/test_abi.py -q -m gpu -x -k "conet or c3 or abi or deferred" 2>&1 | grep -E "passed|failed|error" | tail -3
for rep in 1 2 3; do for v in 1 0; do
  echo "== CDR_CONET_DEFER_FINISH=$v rep $rep: ms_per_step, final loss, abs_total"
  CDR_CONET_DEFER_FINISH=$v python bench.py --workload c3 --steps 400 --warmup 40 --no-cpu-baseline --no-fullsort --detail-file $O/defer_$v.json > /dev/null 2>&1; python -c "import json; d=json.load(open('$O/defer_$v.json')); d=d.get('headline', d); print(d['ms_per_step'], d.get('final_loss'), (d.get('state_checksum') or {}).get('abs_total'))"
done; done
} > $O/ab_defer_finish.txt 2>&1
cat $O/ab_defer_finish.txt
