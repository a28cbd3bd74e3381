"""Two builds of libcdrhip.so on ONE batch: run once per build (`CDR_LIB_PATH=<other .so> python tools/ab_step_builds.py /tmp/tag`), then compare the
saved tables (`np.load(tag + '_U.npy')`) bit for bit and by occurrence count of the differing rows.  Used in round 4 to find that merging the two
duplicate-row applies into one launch left every sum and order alone but changed how hipcc fuses the Adam update (1 ulp on rows with non-zero moments;
identical after the first step, when the moments are still zero)."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recbole_cdr_amd
from recbole_cdr_amd.fused import FusedBPRStep
dev='cuda:0'
torch.manual_seed(0)
nu, ni, D, B = 200000, 50000, 128, 65536
U = (torch.randn(nu, D, device=dev) * 0.1); I = (torch.randn(ni, D, device=dev) * 0.1)
g = torch.Generator(device='cpu').manual_seed(1)
u = torch.randint(0, nu, (B,), generator=g).to(dev); p = torch.randint(0, ni, (B,), generator=g).to(dev); n = torch.randint(0, ni, (B,), generator=g).to(dev)
p[:3000] = 7   # a long segment
st = FusedBPRStep(U, I, B, opt='adam', lr=0.01, reg_weight=0.01)
for _ in range(2):
    st.step(u, p, n)
    print('out', [x.hex() for x in st.out6.cpu().numpy().astype('float32').tolist()[:9]])
torch.cuda.synchronize()
np.save(sys.argv[1] + '_U.npy', U.cpu().numpy()); np.save(sys.argv[1] + '_I.npy', I.cpu().numpy())
np.save(sys.argv[1] + '_ids.npy', torch.stack([u, p, n]).cpu().numpy())
print('done', float(U.double().sum()), float(I.double().sum()))
