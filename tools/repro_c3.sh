# Is `bench.py --workload c3` reproducible from process to process, and is the deferred Adam's trained state the dense sweep's?
# final_loss of the timed epoch + the fp64 sums of the trained state (bench.py `state_checksum`), each variant three times.
ulimit -c 0
O=gpurun_out/repro; mkdir -p $O
run() { tag=$1; shift; for r in 1 2 3; do env "$@" python bench.py --workload ${WL:-c3} --no-cpu-baseline --no-fullsort --steps ${STEPS:-40} --warmup 4 $EXTRA > $O/$tag.$r.json 2> $O/$tag.$r.err; python -c "
import json,hashlib; d=json.loads(open('$O/$tag.$r.json').read().strip().splitlines()[-1]); c=d.get('state_checksum') or {}; print('$tag run $r: final_loss %.10f  ms %.4f  state %s abs_total %s' % (d['final_loss'], d['ms_per_step'], hashlib.md5(json.dumps(c,sort_keys=True).encode()).hexdigest()[:10], c.get('abs_total')))"; done; }
EXTRA="--dense-adam" run dense_adam X=1
EXTRA="" run plain_order CDR_GRAPH_PIPELINE=0
EXTRA="" run unroll1 CDR_GRAPH_UNROLL=1
EXTRA="" run one_ahead CDR_GRAPH_PIPELINE=one_ahead
EXTRA="" run graph_default X=1
