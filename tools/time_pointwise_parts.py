"""Split cabinet_conv1x1 into its three GEMMs (fwd, dx, dw) and time them next to stock F.conv2d fwd / bwd."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from cabinet_amd import _lib
from cabinet_amd.functional import _ptr, _stream_handle, _workspace

LAYERS = [(16, 16, 512), (16, 64, 512), (64, 24, 256), (24, 72, 256), (72, 40, 128), (40, 240, 128), (80, 480, 64),
          (672, 112, 64), (160, 960, 32)]


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


lib = _lib.load()
B = 8
for ci, co, h in LAYERS:
    P = h * h
    x = torch.randn(B, ci, h, h, device="cuda")
    w = torch.randn(co, ci, device="cuda")
    g = torch.randn(B, co, h, h, device="cuda")
    y = torch.empty(B, co, h, h, device="cuda")
    dx, dw = torch.empty_like(x), torch.empty_like(w)
    ws, nb = _workspace(max(lib.cabinet_conv1x1_fwd_workspace_bytes(ci, co),
                            lib.cabinet_conv1x1_bwd_workspace_bytes(B, ci, co, P)), x.device)
    st = _stream_handle(x.device)
    f = timeit(lambda: lib.cabinet_conv1x1_fwd(_ptr(x), _ptr(w), B, ci, co, P, _ptr(y), _ptr(ws), nb, st))
    d = timeit(lambda: lib.cabinet_conv1x1_bwd(_ptr(g), _ptr(x), _ptr(w), B, ci, co, P, _ptr(dx), None, _ptr(ws), nb, st))
    q = timeit(lambda: lib.cabinet_conv1x1_bwd(_ptr(g), _ptr(x), _ptr(w), B, ci, co, P, None, _ptr(dw), _ptr(ws), nb, st))
    xs = x.clone().requires_grad_(True)
    w4 = w.view(co, ci, 1, 1).clone().requires_grad_(True)
    sf = timeit(lambda: F.conv2d(xs, w4))
    ys = F.conv2d(xs, w4)
    sb = timeit(lambda: torch.autograd.grad(ys, (xs, w4), g, retain_graph=True))
    mb = 4.0 * B * P / 1e6
    print(f"{ci:4d}->{co:4d} @{h:3d}: ours fwd {f:7.1f} dx {d:7.1f} dw {q:7.1f} | stock fwd {sf:7.1f} bwd {sb:7.1f} us | "
          f"ideal@5TB/s fwd {mb * (ci + co) / 5:6.1f} dx {mb * (ci + co) / 5:6.1f} dw {mb * (ci + co) / 5:6.1f}")
