#!/bin/bash
# A/B of the deferred Adam's replay kernel at C3: lz_prepare2_kernel (two elements per lane, ring scalars out of LDS; default) against
# lz_prepare1_kernel (CDR_LZ_PREPARE=1).  Output: gpurun_out/r06/ab_lz_prepare.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O
line() { python -c "import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d.get('state_checksum'))"; }
{
cd $R
echo "== tests (deferred Adam, CoNet trainer)"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer_graph.py -q -m gpu -x -k "deferred or lazy or conet or c3" 2>&1 | tail -3
cd /tmp
for rep in 1 2; do
  for v in 0 1; do
    echo "== CDR_LZ_PREPARE=$v rep $rep: ms_per_step, state checksum"
    CDR_LZ_PREPARE=$v python $R/bench.py --workload c3 --steps 400 --warmup 40 --no-cpu-baseline --no-fullsort 2>/dev/null | line
  done
done
for v in 0 1; do
  rm -rf $O/trace_lz_$v
  CDR_LZ_PREPARE=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_lz_$v -o trace -- python $R/bench.py --workload c3 --no-cpu-baseline --no-fullsort --steps 400 --warmup 40 > /dev/null 2>&1
  echo "== kernel stats CDR_LZ_PREPARE=$v"
  f=$(find $O/trace_lz_$v -name '*kernel_stats.csv' | head -1)
  grep -E 'lz_|conet_fb|rank_' $f | cut -c1-200
done
} > $O/ab_lz_prepare.txt 2>&1
tail -60 $O/ab_lz_prepare.txt
