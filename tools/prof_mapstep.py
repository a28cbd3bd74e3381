"""Phase stamps of one 32-id block inside map_pipe_kernel (OVERLAP step, csrc/cdr_mapstep.hip): wall_clock64 ticks (10 ns) of an
MFMA wave and of a row wave at the kernel's MP_STAMP points.  Needs a library built with the stamps compiled in:
    make -C recbole-cdr_amd/csrc clean && make -C recbole-cdr_amd/csrc CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC \
        -munsafe-fp-atomics -Wno-unused-result -DCDR_MAP_PROF"
(then rebuild without the flag: the product library carries no stamps)."""
import os, sys, ctypes, torch, numpy as np
sys.path.insert(0, '/root/repo')
import recbole_cdr_amd
from recbole_cdr_amd import binding as B_, functional as F_
from recbole_cdr_amd.fused import FusedMapStep, RowwiseState, OPT_ADAM
dev='cuda:0'; NU=50_000_001; D=128
S=torch.empty(NU,D,device=dev).normal_(0,1e-3); T=torch.empty(NU,D,device=dev).normal_(0,1e-3)
ss,ts_=RowwiseState(S,OPT_ADAM),RowwiseState(T,OPT_ADAM)
if os.environ.get('MAP', 'linear') == 'linear':
    W=torch.nn.Parameter(torch.randn(D,D,device=dev)*0.05)
    fm=FusedMapStep(S,T,lambda x:F_.linear(x,W,None,B_.ACT_NONE),[W],65536,opt='adam',lr=1e-3,layers=[(W,None,B_.ACT_NONE)],source_state=ss,target_state=ts_)
else:
    W1,b1=torch.nn.Parameter(torch.randn(128,D,device=dev)*0.05),torch.nn.Parameter(torch.zeros(128,device=dev))
    W2,b2=torch.nn.Parameter(torch.randn(D,128,device=dev)*0.05),torch.nn.Parameter(torch.zeros(D,device=dev))
    fm=FusedMapStep(S,T,lambda x:F_.linear(F_.linear(x,W1,b1,B_.ACT_TANH),W2,b2,B_.ACT_NONE),[W1,b1,W2,b2],65536,opt='adam',lr=1e-3,
                    layers=[(W1,b1,B_.ACT_TANH),(W2,b2,B_.ACT_NONE)],source_state=ss,target_state=ts_)
g=torch.Generator(device=dev).manual_seed(0)
perm=torch.randperm(NU-1,device=dev,generator=g)[:65536*4]+1
for i in range(4):
    fm.step(perm[i*65536:(i+1)*65536].view(-1,1).contiguous(),unique=True)
torch.cuda.synchronize()
lib=B_.load()
out=(ctypes.c_longlong*64)()
print('rc',lib.cdr_map_prof_read(out))
a=np.array(list(out),dtype=np.int64)
for base,name in ((0,'consumer'),(32,'producer')):
    v=a[base:base+11]; print(name,[int(x-v[0]) for x in v])
