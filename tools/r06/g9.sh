cd $GRAFT_REPO_ROOT
O=gpurun_out/r06k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_idcount.py -x -q -m gpu 2>&1 | tail -5 | tee $O/tests.txt
cat > /tmp/one.py <<'P'
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch, recbole_cdr_amd
from recbole_cdr_amd.fused import FusedBPRStep
dev=torch.device('cuda:0'); nu,ni,D,B=50_000_000,10_000_000,128,65536
g=torch.Generator(device=dev); g.manual_seed(1)
U=torch.empty(nu,D,device=dev).normal_(0,0.01,generator=g); I=torch.empty(ni,D,device=dev).normal_(0,0.01,generator=g)
st=FusedBPRStep(U,I,B,opt='adam',reg_weight=0.01,id_path=sys.argv[1])
bs=[(torch.randint(1,nu,(B,),device=dev,generator=g),torch.randint(1,ni,(B,),device=dev,generator=g),torch.randint(1,ni,(B,),device=dev,generator=g)) for _ in range(4)]
for i in range(60): st.step(*bs[i%4])
torch.cuda.synchronize()
P
cd /tmp && export TMPDIR=/tmp
for path in count sort; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_$path -o trace -- python /tmp/one.py $path > /dev/null 2>&1
find $GRAFT_REPO_ROOT/$O/trace_$path -name "*kernel_trace.csv" -delete; find $GRAFT_REPO_ROOT/$O/trace_$path -name "*.db" -delete
done
