# Process-to-process reproducibility of: the e2e leg (rowwise trainer epochs + two-stream epochs), and C1 / C2 / C4 with the atomic-free backward.
ulimit -c 0
O=gpurun_out/repro6; mkdir -p $O
for r in 1 2; do
  python bench.py --only-e2e > $O/e2e.$r.json 2> $O/e2e.$r.err
  python -c "
import json; d=json.loads(open('$O/e2e.$r.json').read().strip().splitlines()[-1])['e2e']; print('e2e run $r', d['state_checksum_after_fit'], d['state_checksum_after_two_stream_epochs'], [d['phases'][p]['epoch_loss_sum'] for p in d['phases']])"
done
for wl in c1 c2 c4; do for r in 1 2; do
  CDR_DETERMINISTIC=1 python bench.py --workload $wl --no-cpu-baseline --steps 40 --warmup 4 > $O/$wl.$r.json 2> $O/$wl.$r.err
  python -c "
import json,hashlib; d=json.loads(open('$O/$wl.$r.json').read().strip().splitlines()[-1]); c=d['state_checksum']; print('$wl deterministic run $r: loss %.10f state %s abs_total %s' % (d['final_loss'], hashlib.md5(json.dumps(c,sort_keys=True).encode()).hexdigest()[:10], c['abs_total']))"
done; done
