"""Stem 7x7/2 convolution: cabinet_stem_conv_{fwd,wrw} vs stock MIOpen at the config-3 shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from cabinet_amd.functional import stem_conv


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


conv = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False).cuda()
x = torch.randn(8, 3, 1024, 1024, device="cuda")
g = torch.randn(8, 64, 512, 512, device="cuda")
with torch.no_grad():
    print(f"fwd: ours {timeit(lambda: stem_conv(x, conv)):.0f} us   stock {timeit(lambda: conv(x)):.0f} us")
yo = stem_conv(x, conv)
ys = conv(x)
print(f"wrw: ours {timeit(lambda: torch.autograd.grad(yo, conv.weight, g, retain_graph=True)):.0f} us   "
      f"stock {timeit(lambda: torch.autograd.grad(ys, conv.weight, g, retain_graph=True)):.0f} us")
