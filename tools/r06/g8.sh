cd $GRAFT_REPO_ROOT
O=gpurun_out/r06m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_idcount.py tests/test_gpu_parity.py -x -q -m gpu -k "idcount or count_path or auto_policy or plain_c_consumer or fused_step or sort_ids_stable" 2>&1 | tail -15 | tee $O/tests.txt
timeout 900 python tools/mb_idpath.py > $O/mb_idpath.json 2> $O/mb_idpath.err; echo rc=$?; cat $O/mb_idpath.err | tail -8
