"""dx / dw of the thin pointwise layers separately: cabinet_pwconv_bwd vs aten.convolution_backward (MIOpen)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cabinet_amd import _lib
from cabinet_amd.functional import _ptr, _stream_handle, _workspace

LAYERS = [(16, 16, 512), (16, 64, 512), (64, 24, 256), (24, 72, 256), (72, 24, 256), (72, 40, 128)]


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
    w = torch.randn(co, ci, 1, 1, device="cuda")
    g = torch.randn(B, co, h, h, device="cuda")
    dx, dw = torch.empty_like(x), torch.empty(co, ci, device="cuda")
    ws, nb = _workspace(lib.cabinet_pwconv_bwd_workspace_bytes(B, ci, co, P), x.device)
    st = _stream_handle(x.device)
    d1 = timeit(lambda: lib.cabinet_pwconv_bwd(_ptr(g), _ptr(x), _ptr(w), B, ci, co, P, _ptr(dx), None, _ptr(ws), nb, st))
    w1 = timeit(lambda: lib.cabinet_pwconv_bwd(_ptr(g), _ptr(x), _ptr(w), B, ci, co, P, None, _ptr(dw), _ptr(ws), nb, st))
    cb = torch.ops.aten.convolution_backward
    d0 = timeit(lambda: cb(g, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False]))
    w0 = timeit(lambda: cb(g, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False]))
    print(f"{ci:4d}->{co:4d} @{h:3d}: dx ours {d1:7.1f} stock {d0:7.1f} | dw ours {w1:7.1f} stock {w0:7.1f} us")
