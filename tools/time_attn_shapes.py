import sys; sys.path.insert(0,'.')
import torch
from cabinet_amd import functional as Fh
def t(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/it*1e3
for (B,Kc,Vc,n) in [(2,128,128,2048),(8,128,128,1024),(1,256,128,2048),(4,128,128,256),(1,128,128,8704)]:
    g=torch.Generator().manual_seed(0)
    q=torch.randn(B,Kc,n,generator=g).relu().cuda(); k=torch.randn(B,Kc,n,generator=g).cuda(); v=torch.randn(B,Vc,n,generator=g).cuda(); d=torch.randn(B,Vc,n,generator=g).cuda()
    ctx,lse=Fh.attn_fwd_hip(q,k,v,Kc**-0.5)
    f=t(lambda: Fh.attn_fwd_hip(q,k,v,Kc**-0.5)); bw=t(lambda: Fh.attn_bwd_hip(d,q,k,v,ctx,lse,Kc**-0.5))
    fl=2.0*B*n*n*(Kc+Vc); flb=2.0*B*n*n*(3*Kc+2*Vc)
    print(f"B={B} Kc={Kc} Vc={Vc} n={n}: fwd {f:8.1f} us {fl/f/1e6:6.1f} TF/s ({fl/f/1e6/157.3:.2f}) | bwd {bw:8.1f} us {flb/bw/1e6:6.1f} TF/s ({flb/bw/1e6/157.3:.2f})")
