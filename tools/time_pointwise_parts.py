"""Split cabinet_conv1x1 into its three GEMMs (fwd, dx, dw) and time them next to stock F.conv2d fwd / bwd."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from cabinet_amd import _lib
from cabinet_amd.functional import _ptr, _stream_handle, _workspace

LAYERS = [(72, 40, 128), (40, 120, 128), (120, 40, 128), (40, 240, 128), (240, 80, 64), (80, 200, 64), (200, 80, 64),
          (80, 184, 64), (184, 80, 64), (80, 480, 64), (480, 112, 64), (112, 672, 64), (672, 112, 64), (672, 160, 32),
          (160, 960, 32), (960, 160, 32)]


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
    w4 = w.view(co, ci, 1, 1)
    cb = torch.ops.aten.convolution_backward
    sf = timeit(lambda: F.conv2d(x, w4))
    sdx = timeit(lambda: cb(g, x, w4, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False]))
    sdw = timeit(lambda: cb(g, x, w4, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False]))
    sb = sdx + sdw
    mb = 4.0 * B * P / 1e6
    gf = 2.0 * B * P * ci * co / 1e9
    print(f"{ci:4d}->{co:4d} @{h:3d}: ours fwd {f:6.1f} dx {d:6.1f} dw {q:6.1f} | stock fwd {sf:6.1f} dx {sdx:6.1f} dw {sdw:6.1f} us | "
          f"HBM@5TB/s {mb * (ci + co) / 5:5.1f} us, MFMA@100TF {gf * 10:5.1f} us per product")
