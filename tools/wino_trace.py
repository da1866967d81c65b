#!/usr/bin/env python3
"""Where a chunk of K11's forward kernel spends its cycles: build with CABINET_EXTRA_HIPCC_FLAGS=-DWN_TRACE, then
    python tools/wino_trace.py [conva|b1|out]        (on the GPU box; BASELINE config 3 shapes)
prints, for waves 0 and 4 of workgroup 100 (the two waves of one SIMD), the cycles of every quarter of its first chunks.
A quarter holds 8 MFMAs of the wave = 512 cycles of the SIMD's matrix pipe, 1024 with the partner wave's."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from cabinet_amd import _lib, functional as Fn  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "out"
B, C0, C1, K, H, W = {"conva": (8, 960, 0, 256, 32, 32), "b1": (8, 960, 256, 256, 32, 32), "out": (8, 256, 0, 256, 128, 128)}[which]
x0 = torch.randn(B, C0, H, W, device="cuda")
x1 = torch.randn(B, C1, H, W, device="cuda") if C1 else None
w = torch.randn(K, C0 + C1, 3, 3, device="cuda") * 0.02
for _ in range(3):
    Fn.conv3x3_fwd_hip(x0, x1, w)
torch.cuda.synchronize()
lib = _lib.load()
n = 2 * 32 * 8
buf = (ctypes.c_ulonglong * n)()
lib.cabinet_debug_wn_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.cabinet_debug_wn_trace(buf, n) == 0
names = ["Q0 (U loads, LDS reads)", "Q1 (32 adds, 16 loads)", "Q2 (8 LDS stores)", "barrier", "Q3 (LDS reads)", "-> next Q0"]
print(f"K11 forward, shape {which}: cycles per quarter of a chunk (a quarter = 8 MFMAs of the wave = 512 cycles of the pipe, 1024 with the partner's)")
for wv in range(2):
    print(f"wave {4 * wv}: chunk " + " ".join(f"{x:>24s}" for x in names) + "   chunk total")
    for c in range(16):
        t = [buf[(wv * 32 + c) * 8 + i] for i in range(6)]
        nxt = buf[(wv * 32 + c + 1) * 8]
        if t[0] == 0 or nxt == 0:
            continue
        d = [t[i + 1] - t[i] for i in range(5)] + [nxt - t[5]]
        print(f"        {c:5d} " + " ".join(f"{x:24d}" for x in d) + f"   {nxt - t[0]:8d}")
