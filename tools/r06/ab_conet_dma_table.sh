#!/bin/bash
# A/B of conet_fb_kernel's weight staging: DMA sources from a host-built table (default) against computed per lane (CDR_CONET_NO_DMA_TABLE=1).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
{
python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer_graph.py -q -m gpu -x -k "conet or c3" 2>&1 | grep -E "passed|failed" | tail -2
for i in 1 2 3; do for v in "" 1; do echo -n "no_table=${v:-0}: "; env ${v:+CDR_CONET_NO_DMA_TABLE=1} python tools/mb_conet.py 2>/dev/null | grep -E "conet_fb" | tr '\n' ' '; env ${v:+CDR_CONET_NO_DMA_TABLE=1} python bench.py --workload c3 --steps 480 --warmup 32 --no-cpu-baseline --no-fullsort --detail-file gpurun_out/r06/dmatab.json > /dev/null 2>&1; python -c "import json; d=json.load(open('gpurun_out/r06/dmatab.json')); print(d['ms_per_step'], d['final_loss'], d['state_checksum']['abs_total'])"; done; done
} 2>&1 | tee gpurun_out/r06/ab_conet_dma_table.txt
