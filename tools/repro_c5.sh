# The headline step from process to process: fp64 sums of the four trained tables (bench.py `state_checksum`), two streams and one.
ulimit -c 0
O=gpurun_out/repro5; mkdir -p $O
for tag in two_a two_b one_a; do
  extra=""; [ $tag = one_a ] && extra="--single-stream"
  python bench.py --headline-only --steps 10 --warmup 2 $extra > $O/$tag.json 2> $O/$tag.err
  python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['final_loss'], d.get('state_checksum'))"
done
