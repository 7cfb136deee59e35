import torch, time
dev='cuda'
def bench(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
torch.backends.cuda.matmul.allow_tf32=False
for E in (25600, 10000, 3840, 102400, 15104):
    A=torch.randn(E,256,device=dev); W=torch.randn(256,256,device=dev); dY=torch.randn(E,256,device=dev); G=torch.zeros(256,256,device=dev)
    t1=bench(lambda: torch.mm(A,W.t()))
    t2=bench(lambda: torch.mm(dY,W))
    t3=bench(lambda: G.addmm_(dY.t(),A))
    fl=2*E*256*256
    print(f'E={E}: fwd {t1:.1f}us {fl/t1/1e6:.1f}TF  dgrad {t2:.1f}us {fl/t2/1e6:.1f}TF  wgrad {t3:.1f}us {fl/t3/1e6:.1f}TF')
A=torch.randn(3840,256,device=dev); W2=torch.randn(512,256,device=dev)
print('P|Q N=512', bench(lambda: torch.mm(A,W2.t())))
