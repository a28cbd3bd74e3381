"""C5 table sizes: one domain step of FusedBPRStep, per-kernel HIP-event times (library brackets) and the step's wall time.
usage: python tools/mb_step.py [users] [items] [B] [D] [zipf]   env CDR_FUSE_SINGLES=0 -> the two-pass path"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recbole_cdr_amd  # noqa: F401
from recbole_cdr_amd import binding as B_
from recbole_cdr_amd.fused import FusedBPRStep
nu = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_001
ni = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_001
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
D = int(sys.argv[4]) if len(sys.argv) > 4 else 128
zipf = len(sys.argv) > 5 and sys.argv[5] == 'zipf'
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(2022)
U = torch.empty(nu, D, device=dev).normal_(0, 0.01, generator=g)
I = torch.empty(ni, D, device=dev).normal_(0, 0.01, generator=g)
st = FusedBPRStep(U, I, B, opt='adam', reg_weight=float(os.environ.get('CDR_MB_REG', '0.01')))
half = ni // 2
def items():
    if not zipf:
        return torch.randint(1 + half, ni, (B,), device=dev, generator=g)
    r = torch.rand(B, device=dev, generator=g).double()
    # Zipf(1.05) over `half` ranks by inverse CDF of the continuous approximation
    a = 1.05
    x = ((half ** (1 - a) - 1) * r + 1) ** (1 / (1 - a))
    return (1 + half + (x.long().clamp_(1, half) - 1))
bs = [(torch.randint(1, nu, (B,), device=dev, generator=g), items(), torch.randint(1 + half, ni, (B,), device=dev, generator=g)) for _ in range(4)]
for i in range(5):
    st.step(*bs[i % 4])
torch.cuda.synchronize()
B_.timing_enable(dev, 4096)
N = 20
t0 = time.perf_counter()
for i in range(N):
    st.step(*bs[i % 4])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N * 1e3
kt = {}
for nm, ms in B_.timing_collect(dev):
    kt.setdefault(nm, []).append(ms)
print('fuse_singles=%s UN=%s zipf=%s: %.3f ms per domain step of %d triples (%.3e triples/s; 9216 B/triple -> %.2f TB/s)' % (
    st.fuse_singles, os.environ.get('CDR_FWD_APPLY_UN', '1'), zipf, dt, B, B / dt * 1e3, 9216.0 * B / dt / 1e9))
for k, v in kt.items():
    v = sorted(v)
    print('  %-32s n=%3d median %.4f ms  min %.4f  max %.4f' % (k, len(v), v[len(v) // 2], v[0], v[-1]))
if st.fuse_singles:
    torch.cuda.synchronize()
    c = st.heads[:2].tolist()
    f = st.flags[:4 * B].view(B, 4).float().mean(0).tolist()
    print('  single fractions u/p/n: %.4f %.4f %.4f; duplicate segments users %d items %d' % (f[0], f[1], f[2], c[0], c[1]))
print('  loss %.6f' % float(st.out6[0]))
