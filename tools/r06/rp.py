import torch, time
dev='cuda:0'
g=torch.Generator(device=dev); g.manual_seed(1)
n=9_437_184
a=torch.arange(n,device=dev); b=torch.arange(n,device=dev)
def t(fn,reps=10):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps*1e3
print('randperm', t(lambda: torch.randperm(n,generator=g,device=dev)))
p=torch.randperm(n,generator=g,device=dev)
print('permute 2 cols in place', t(lambda: (a.copy_(a[p]), b.copy_(b[p]))))
print('rand keys + sort', t(lambda: torch.sort(torch.randint(0,2**31-1,(n,),device=dev,generator=g,dtype=torch.int32))))
print('argsort of rand float', t(lambda: torch.argsort(torch.rand(n,device=dev,generator=g))))
