# Every C5 leg of the default bench run (headline, OVERLAP linear + MLP, k-major, Zipf / B=65536 grid incl. hipGraph replays, small batch) twice:
# the tables' fp64 sums after all of them.
ulimit -c 0
O=gpurun_out/repro8; mkdir -p $O
for r in 1 2; do
  python bench.py --no-cpu-baseline --no-config-legs --no-e2e --steps 10 --warmup 2 > $O/legs.$r.json 2> $O/legs.$r.err
  python -c "
import json; d=json.loads(open('$O/legs.$r.json').read().strip().splitlines()[-1]); print('run $r', d['bench_wall_s'], d['state_checksum'], d['state_checksum_after_all_legs'], d.get('leg_errors'))"
done
