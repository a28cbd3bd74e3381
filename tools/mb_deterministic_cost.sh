# Cost of the atomic-free dense backward (functional.set_deterministic / CDR_DETERMINISTIC=1) on the configurations it covers:
# `bench.py --workload c1|c2` through CrossDomainTrainer.fit with and without it.  Output: gpurun_out/det/summary.txt
ulimit -c 0
O=gpurun_out/det; mkdir -p $O
for wl in c1 c2 c4; do
  for d in 0 1; do
    CDR_DETERMINISTIC=$d python bench.py --workload $wl --no-cpu-baseline --steps 200 --warmup 20 > $O/${wl}_det$d.json 2> $O/${wl}_det$d.err || echo "$wl det=$d rc=$?"
  done
done
python - <<'PY' | tee gpurun_out/det/summary.txt
import json
print('atomic-free dense backward (CDR_DETERMINISTIC=1: cdr_ordered_bwd for lists <= 8,192 since round 5, the sorted form beyond) vs the default float-atomic scatter; bench.py --workload cN through CrossDomainTrainer.fit, 200 steps')
for wl in ("c1", "c2", "c4"):
    r = {}
    for d in (0, 1):
        try:
            r[d] = json.loads(open('gpurun_out/det/%s_det%d.json' % (wl, d)).read().strip().splitlines()[-1])
        except Exception as e:
            r[d] = {'error': repr(e)}
    for d in (0, 1):
        x = r[d]
        print('%s deterministic=%d: ms_per_step %s  value %s  flag_in_line %s' % (
            wl, d, x.get('ms_per_step'), x.get('value'), x.get('deterministic_backward')))
PY
