cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_trainer_graph.py tests/test_gpu_models5.py -x -q -m gpu 2>&1 | tail -4
for w in c1 c2 c4; do python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$w', round(d['ms_per_step'],5), d.get('final_loss'))"; done
