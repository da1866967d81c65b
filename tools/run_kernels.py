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
torch.cuda.synchronize()
print("done")
