cd $GRAFT_REPO_ROOT
for i in 1 2; do for un in 1 2; do CDR_FWD_APPLY_UN=$un python bench.py --headline-only --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('UN=$un', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4))"; done; done
