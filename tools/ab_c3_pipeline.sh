# A/B of the unrolled CoNet step's launch order (bench.py --workload c3): plain | one batch ahead on the side stream | two batches ahead
ulimit -c 0
O=gpurun_out/c3p; mkdir -p $O
:
for v in "1 8" "two_ahead 8" "0 8" "1 16" "two_ahead 16"; do
  set -- $v
  CDR_GRAPH_PIPELINE=$1 CDR_GRAPH_UNROLL=$2 python bench.py --workload c3 --no-cpu-baseline --no-fullsort --steps 400 --warmup 40 > $O/p$1_u$2.json 2> $O/p$1_u$2.err || echo "rc=$? for $v"
  python - <<PY
import json
try:
    d=json.loads(open('$O/p$1_u$2.json').read().strip().splitlines()[-1]); print('pipeline=$1 unroll=$2: %.4f ms per step, loss %.9f, %s' % (d['ms_per_step'], d['final_loss'], d['config']['trainer_steps']))
except Exception as e: print('pipeline=$1 unroll=$2: ERR', e); print(open('$O/p$1_u$2.err').read()[-1500:])
PY
done
