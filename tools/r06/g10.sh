cd $GRAFT_REPO_ROOT
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
for i in range(20): st.step(*bs[i%4])
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(100): st.step(*bs[i%4])
e1.record(); torch.cuda.synchronize()
print(sys.argv[1], os.environ.get('CDR_DBG_COUNT'), round(e0.elapsed_time(e1)/100,4))
P
for d in 0; do CDR_DBG_COUNT=$d python /tmp/one.py count; done
python /tmp/one.py sort
