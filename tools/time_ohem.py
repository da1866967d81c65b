import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cabinet_amd.functional import ohem_up_fwd_hip, ohem_up_bwd_hip
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/it*1e3
B,C,Hl,Wl,H,W=8,8,128,128,1024,1024
g=torch.Generator().manual_seed(0)
low=(torch.randn(B,C,Hl,Wl,generator=g)*2).cuda(); lab=torch.randint(0,C,(B,H,W),generator=g).cuda()
loss_px,stats=ohem_up_fwd_hip(low,lab,(H,W),0.7,255)
print("fwd us", t(lambda: ohem_up_fwd_hip(low,lab,(H,W),0.7,255)))
print("bwd us", t(lambda: ohem_up_bwd_hip(low,lab,loss_px,(H,W),0.7,255,1e-6)))
import torch.nn.functional as F
x=low.clone().requires_grad_(True)
def ref():
    up=F.interpolate(x,size=(H,W),mode="bilinear",align_corners=False)
    l=F.cross_entropy(up,lab,reduction="none")
    m=(l>0.7)
    ((l*m).sum()/m.sum()).backward()
print("unfused fwd+bwd us", t(ref))
