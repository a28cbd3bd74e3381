cd $GRAFT_REPO_ROOT
O=gpurun_out/r06o; mkdir -p $O
cat > /tmp/zipf.py <<'P'
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch, recbole_cdr_amd
from recbole_cdr_amd.fused import FusedBPRStep
dev=torch.device('cuda:0'); nu,ni,D,B=50_000_000,10_000_000,128,65536
g=torch.Generator(device=dev); g.manual_seed(1)
U=torch.empty(nu,D,device=dev).normal_(0,0.01,generator=g); I=torch.empty(ni,D,device=dev).normal_(0,0.01,generator=g)
def zipf(n):
    r=torch.rand(n,device=dev,generator=g,dtype=torch.float64); a=1.05
    x=((float(ni-1)**(1-a)-1)*r+1)**(1/(1-a)); return x.long().clamp_(1,ni-1)
st=FusedBPRStep(U,I,B,opt='adam',reg_weight=0.01,id_path='sort')
bs=[(torch.randint(1,nu,(B,),device=dev,generator=g),zipf(B),torch.randint(1,ni,(B,),device=dev,generator=g)) for _ in range(4)]
for i in range(60): st.step(*bs[i%4])
torch.cuda.synchronize()
P
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_zipf -o trace -- python /tmp/zipf.py > /dev/null 2>&1
find $GRAFT_REPO_ROOT/$O -name "*kernel_trace.csv" -delete; find $GRAFT_REPO_ROOT/$O -name "*.db" -delete
