"""cdr_sort_ids_small alone: microseconds per launch for the list shapes the small-batch steps use."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recbole_cdr_amd  # noqa
from recbole_cdr_amd import binding as B_
dev = 'cuda:0'
g = torch.Generator(device=dev).manual_seed(0)
for sizes in ([(100, 0)], [(2048, 0), (2048, 2048)], [(512, 0), (512, 2048)], [(4096, 0), (4096, 4096)], [(8190, 0), (8190, 0)], [(4095, 4095), (4095, 4095)], [(8192, 0), (8192, 8192)]):
    a = [torch.randint(0, int(os.environ.get('IDMAX', 50_000_000)), (n0,), device=dev, generator=g) for n0, _ in sizes]
    b = [torch.randint(0, int(os.environ.get('IDMAX', 10_000_000)), (n1,), device=dev, generator=g) if n1 else None for _, n1 in sizes]
    offs, tot = [], 0
    for n0, n1 in sizes:
        offs.append(tot); tot += n0 + n1
    keys = torch.empty(tot, device=dev, dtype=torch.int32); perm = torch.empty(tot, device=dev, dtype=torch.int32)
    rank = torch.zeros(tot, device=dev, dtype=torch.int32)
    ns = len(sizes)
    args = (B_.stream(), ns, (ctypes.c_void_p * ns)(*[t.data_ptr() for t in a]), (ctypes.c_int64 * ns)(*[n0 for n0, _ in sizes]),
            (ctypes.c_void_p * ns)(*[t.data_ptr() if t is not None else None for t in b]), (ctypes.c_int64 * ns)(*[n1 for _, n1 in sizes]),
            (ctypes.c_int64 * ns)(*offs), B_.raw(keys), B_.raw(perm), B_.raw(rank), int(os.environ.get('MAXID', 0)))
    for _ in range(5):
        B_.call('cdr_sort_ids_small', *args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(200):
        B_.call('cdr_sort_ids_small', *args)
    e1.record(); torch.cuda.synchronize()
    print(sizes, '%.1f us per launch' % (e0.elapsed_time(e1) / 200 * 1e3))
