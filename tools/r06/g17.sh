cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_idcount.py -x -q -m gpu -k "fused or long or zipf or sharded_step_world1 or idcount or count_path or rowwise or kmajor or edge_wide" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
python tools/mb_idpath.py 2>&1 >/dev/null | grep -E "zipf" | cut -c1-140
