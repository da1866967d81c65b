"""Thin pointwise layers of the backbone (config 3): cabinet_pwconv vs stock MIOpen, fwd and bwd."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from cabinet_amd.functional import pwconv

LAYERS = [(16, 16, 512), (16, 64, 512), (64, 24, 256), (24, 72, 256), (72, 24, 256), (72, 40, 128), (40, 120, 128),
          (120, 40, 128)]


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


tot = [0, 0, 0, 0]
for ci, co, h in LAYERS:
    conv = torch.nn.Conv2d(ci, co, 1, bias=False).cuda()
    x = torch.randn(8, ci, h, h, device="cuda", requires_grad=True)
    g = torch.randn(8, co, h, h, device="cuda")
    with torch.no_grad():
        f1, f0 = timeit(lambda: pwconv(x, conv)), timeit(lambda: conv(x))
    y1, y0 = pwconv(x, conv), conv(x)
    b1 = timeit(lambda: torch.autograd.grad(y1, (x, conv.weight), g, retain_graph=True))
    b0 = timeit(lambda: torch.autograd.grad(y0, (x, conv.weight), g, retain_graph=True))
    mb = 4.0 * 8 * h * h * (ci + co) / 1e6
    print(f"{ci:4d}->{co:4d} @{h:3d}: fwd ours {f1:7.1f} stock {f0:7.1f} | bwd ours {b1:7.1f} stock {b0:7.1f} us | "
          f"HBM-ideal fwd {mb / 5:6.1f} bwd {2 * mb / 5:6.1f}")
    for i, v in enumerate((f1, f0, b1, b0)):
        tot[i] += v
print(f"total: fwd ours {tot[0] / 1e3:.2f} stock {tot[1] / 1e3:.2f} ms | bwd ours {tot[2] / 1e3:.2f} stock {tot[3] / 1e3:.2f} ms")
