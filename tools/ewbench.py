import sys, torch
sys.path.insert(0,'/root/repo')
from neuralsvb_amd import kernels as K
dev=torch.device('cuda:0')
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters*1e-3
for T in (1124, 281):
    B,C=32,192
    xin=torch.randn(B,2*C,T,device=dev); G=torch.randn(B,2*C*4,T,device=dev); dacts=torch.randn(B,C,T,device=dev)
    x=torch.randn(B,C,T,device=dev); rs=torch.randn(B,2*C,T,device=dev); out=torch.randn(B,C,T,device=dev); mask=torch.ones(B,T,device=dev)
    for q in (False, True):
        t=timeit(lambda: K.wn_gate_fwd(xin,G,0,want_q=q)); print(f"T{T} gate_fwd q={q}: {t*1e6:.1f} us  {5*C*B*T*4/t/1e12:.2f} TB/s(fp32 bytes)")
        t=timeit(lambda: K.wn_gate_bwd(xin,G,dacts,0,want_q=q)); print(f"T{T} gate_bwd q={q}: {t*1e6:.1f} us  {7*C*B*T*4/t/1e12:.2f} TB/s")
        t=timeit(lambda: K.wn_res_skip(x,rs,mask,out,False,want_q=q)); print(f"T{T} res_skip q={q}: {t*1e6:.1f} us  {6*C*B*T*4/t/1e12:.2f} TB/s")
        t=timeit(lambda: K.wn_res_skip_bwd(x,out,mask,want_q=q)); print(f"T{T} res_skip_bwd q={q}: {t*1e6:.1f} us  {5*C*B*T*4/t/1e12:.2f} TB/s")
