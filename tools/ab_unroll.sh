# steps per graph launch: 8 (default) vs 16, C1-C4 through CrossDomainTrainer.fit
ulimit -c 0
O=gpurun_out/unroll; mkdir -p $O
for wl in c1 c2 c3 c4; do for u in 8 16 32 8 16 32; do
  CDR_GRAPH_UNROLL=$u python bench.py --workload $wl --no-cpu-baseline --no-fullsort --steps 480 --warmup 20 > $O/$wl.$u.json 2> $O/$wl.$u.err
  python -c "
import json; d=json.loads(open('$O/$wl.$u.json').read().strip().splitlines()[-1]); print('$wl unroll $u: %.4f ms per step  %s' % (d['ms_per_step'], d['config']['trainer_steps']['replayed']))"
done; done
