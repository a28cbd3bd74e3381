ulimit -c 0
O=gpurun_out/repro7; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -k "deterministic" 2>&1 | tail -3
for r in 1 2 3; do
  CDR_DETERMINISTIC=1 python bench.py --workload c4 --no-cpu-baseline --steps 40 --warmup 4 > $O/c4.$r.json 2> $O/c4.$r.err
  python -c "
import json,hashlib; d=json.loads(open('$O/c4.$r.json').read().strip().splitlines()[-1]); c=d['state_checksum']; print('c4 deterministic run $r: ms %.4f loss %.10f state %s abs_total %s' % (d['ms_per_step'], d['final_loss'], hashlib.md5(json.dumps(c,sort_keys=True).encode()).hexdigest()[:10], c['abs_total']))"
done
