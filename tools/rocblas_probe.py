import torch, numpy as np
def timeit(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
M,D=558771,200
x=torch.randn(M,D,device="cuda"); x2=torch.randn(M,2*D,device="cuda"); dz=torch.randn(M,D,device="cuda")
W=torch.randn(D,2*D,device="cuda"); b=torch.randn(D,device="cuda"); W1=torch.randn(D,D,device="cuda")
for name,fn,fl in [
 ("addmm [M,400]x[400,200]", lambda: torch.addmm(b,x2,W.t()), 2*M*400*200),
 ("mm dz[M,200] x W[200,400]", lambda: torch.mm(dz,W), 2*M*400*200),
 ("mm dz.T[200,M] x x2[M,400]", lambda: torch.mm(dz.t(),x2), 2*M*400*200),
 ("mm dz.T[200,M] x x[M,200]", lambda: torch.mm(dz.t(),x), 2*M*200*200),
 ("mm dz[M,200] x W1[200,200]", lambda: torch.mm(dz,W1), 2*M*200*200),
 ("cat [M,200]+[M,200]", lambda: torch.cat([x,dz],1), 0),
]:
    ms=timeit(fn); print(f"{name:32s} {ms:8.3f} ms  {fl/ms/1e9:8.1f} TF/s")
