#!/usr/bin/env python3
"""Launch each hand-written kernel group at BASELINE config-3 shapes a few times (for rocprofv3 passes)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from cabinet_amd import functional as Fh  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
which = sys.argv[2] if len(sys.argv) > 2 else "all"
B = int(os.environ.get("CAB_B", "8"))
size = int(os.environ.get("CAB_SIZE", "1024"))
dev = "cuda"
g = torch.Generator().manual_seed(3)
Kc = Vc = 128
n = (size // 32) ** 2
h = w = size // 8
q = torch.randn(B, Kc, n, generator=g).relu().to(dev)
k = torch.randn(B, Kc, n, generator=g).to(dev)
v = torch.randn(B, Vc, n, generator=g).to(dev)
dctx = torch.randn(B, Vc, n, generator=g).to(dev)
scale = Kc ** -0.5
if which in ("all", "attn"):
    for _ in range(iters):
        ctx, lse = Fh.attn_fwd_hip(q, k, v, scale)
    for _ in range(iters):
        Fh.attn_bwd_hip(dctx, q, k, v, ctx, lse, scale)
if which in ("all", "ffm"):
    Cs, Cc, Co, Cm = 128, 256, 256, 64
    fsp = torch.randn(B, Cs, h, w, generator=g).to(dev)
    fcp = torch.randn(B, Cc, h, w, generator=g).to(dev)
    wb = (torch.randn(Co, Cs + Cc, generator=g) * 0.07).to(dev)
    w1 = (torch.randn(Cm, Co, generator=g) * 0.1).to(dev)
    w2 = (torch.randn(Co, Cm, generator=g) * 0.1).to(dev)
    bw, bb = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
    rm, rv = torch.zeros(Co, device=dev), torch.ones(Co, device=dev)
    dout = torch.randn(B, Co, h, w, generator=g).to(dev)
    for _ in range(iters):
        o, z, mean, invstd, pooled, gate = Fh.ffm_fwd_hip(fsp, fcp, wb, bw, bb, rm, rv, w1, w2, True, 0.1, 1e-5)
    for _ in range(iters):
        Fh.ffm_bwd_hip(dout, fsp, fcp, wb, bw, bb, w1, w2, z, mean, invstd, pooled, gate, True)
if which in ("all", "ffm_up"):
    Cs, Cc, Co, Cm = 128, 256, 256, 64
    fsp = torch.randn(B, Cs, h, w, generator=g).to(dev)
    low = torch.randn(B, Cc, size // 32, size // 32, generator=g).to(dev)
    wb = (torch.randn(Co, Cs + Cc, generator=g) * 0.07).to(dev)
    w1 = (torch.randn(Cm, Co, generator=g) * 0.1).to(dev)
    w2 = (torch.randn(Co, Cm, generator=g) * 0.1).to(dev)
    bw, bb = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
    rm, rv = torch.zeros(Co, device=dev), torch.ones(Co, device=dev)
    dout = torch.randn(B, Co, h, w, generator=g).to(dev)
    for _ in range(iters):
        o, z, mean, invstd, pooled, gate = Fh.ffm_up_fwd_hip(fsp, low, wb, bw, bb, rm, rv, w1, w2, True, 0.1, 1e-5)
    for _ in range(iters):
        Fh.ffm_up_bwd_hip(dout, fsp, low, wb, bw, bb, w1, w2, z, mean, invstd, pooled, gate, True)
if which in ("all", "new", "bn_act"):
    # K7 at the largest plane of the model (sb.conv1 / features.2): 8 x 64 x 512 x 512
    xb = torch.randn(B, 64, size // 2, size // 2, generator=g).to(dev).requires_grad_(True)
    gb = torch.randn(B, 64, size // 2, size // 2, generator=g).to(dev)
    bnw, bnb = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    brm, brv = torch.zeros(64, device=dev), torch.ones(64, device=dev)
    for _ in range(iters):
        yb = Fh._BnAct.apply(xb, bnw, bnb, brm, brv, 2, True, 0.1, 1e-5)
        torch.autograd.grad(yb, xb, gb)
    del xb, gb, yb
if which in ("all", "new", "dwconv"):
    # K8 on features.2 (64 ch, 512 -> 256, 3x3 stride 2), plain and with the BatchNorm folded in, and a 5x5 stride-1 layer
    import torch.nn as nn

    for ch, hw, k, st in ((64, size // 2, 3, 2), (120, size // 8, 5, 1)):
        conv = nn.Conv2d(ch, ch, k, st, k // 2, groups=ch, bias=False).to(dev)
        bn = nn.BatchNorm2d(ch).to(dev).train()
        xd = torch.randn(B, ch, hw, hw, generator=g).to(dev).requires_grad_(True)
        for _ in range(iters):
            y = Fh.dwconv(xd, conv)
            torch.autograd.grad(y, (xd, conv.weight), torch.ones_like(y))
        for _ in range(iters):
            y = Fh.bn_act_dwconv(xd, bn, "hardswish", conv)
            torch.autograd.grad(y, (xd, conv.weight, bn.weight), torch.ones_like(y))
        del xd, y
if which in ("all", "new", "ohem"):
    lowl = torch.randn(B, 8, size // 8, size // 8, generator=g).to(dev)
    lab = torch.randint(0, 8, (B, size, size), generator=g).to(dev)
    for _ in range(iters):
        loss_px, _ = Fh.ohem_up_fwd_hip(lowl, lab, (size, size), 0.7, 255)
        Fh.ohem_up_bwd_hip(lowl, lab, loss_px, (size, size), 0.7, 255, 1e-6)
if which in ("all", "new", "cab"):
    from cabinet_amd.models.cab import ContextAggregationBlock

    cab = ContextAggregationBlock(256, 128).to(dev).train()
    xc = torch.randn(B, 256, size // 32, size // 32, generator=g).to(dev).requires_grad_(True)
    for _ in range(iters):
        yl = cab.local_attn(xc)
        torch.autograd.grad(yl, xc, torch.ones_like(yl))
        q3 = Fh.cab_qkv(xc, cab.global_attn)
        torch.autograd.grad(q3, xc, [torch.ones_like(t) for t in q3])
torch.cuda.synchronize()
print("done")
