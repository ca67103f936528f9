import sys, torch
sys.path.insert(0,'/root/repo')
from neuralsvb_amd import kernels as K
dev=torch.device('cuda:0')
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters*1e-3
for (B,ca,cb,T,k) in [(32,384,192,1124,1),(32,384,192,281,1),(32,1536,256,1124,1),(32,3072,256,281,1),(32,256,768,1124,1),(32,256,256,1124,1),(32,1024,256,562,1),(32,256,1024,562,1)]:
    a=torch.randn(B,ca,T,device=dev); b=torch.randn(B,cb,T,device=dev)
    t=timeit(lambda: K.conv1d_wgrad(a,b,k,1,0,1,1,bf16x3=True))
    print(f"wgrad A{ca} B{cb} T{T} k{k}: {t*1e6:.1f} us {2.0*B*ca*cb*T*k/t/1e12:.1f} TF")
