#!/usr/bin/env python3
"""Where a chunk of the fused FFM backward kernel spends its cycles: build with CABINET_EXTRA_HIPCC_FLAGS=-DXW_TRACE, then
    python tools/xw_trace.py            (on the GPU box; BASELINE config 3 operand shapes)
prints, for waves 0 and 4 of workgroup 100, the cycle count of every phase of every chunk of its run."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from cabinet_amd import _lib, functional as Fn  # noqa: E402

B, H, W, Hl, Wl = 8, 128, 128, 32, 32
gen = torch.Generator().manual_seed(0)
Cs, Cc, Co, Cm = 128, 256, 256, 64
fsp = torch.randn(B, Cs, H, W, generator=gen).cuda()
low = torch.randn(B, Cc, Hl, Wl, generator=gen).cuda()
wb = (torch.randn(Co, Cs + Cc, generator=gen) * 0.07).cuda()
w1, w2 = (torch.randn(Cm, Co, generator=gen) * 0.1).cuda(), (torch.randn(Co, Cm, generator=gen) * 0.1).cuda()
g = torch.randn(B, Co, H, W, generator=gen).cuda()
bw, bb = torch.rand(Co, generator=gen).cuda() + 0.5, torch.rand(Co, generator=gen).cuda() - 0.5
rm, rv = torch.zeros(Co).cuda(), torch.ones(Co).cuda()
out, z, mean, invstd, pooled, gate = Fn.ffm_up_fwd_hip(fsp, low, wb, bw, bb, rm, rv, w1, w2, True, 0.1, 1e-5)
args = (g, fsp, low, wb, bw, bb, w1, w2, z, mean, invstd, pooled, gate, True)
fwd = len(sys.argv) > 1 and sys.argv[1] == "fwd"   # the forward kernel (ffm_fwd_fused.hip, -DFZ_TRACE) instead
for _ in range(3):
    if fwd:
        Fn.ffm_up_fwd_hip(fsp, low, wb, bw, bb, rm, rv, w1, w2, True, 0.1, 1e-5)
    else:
        Fn.ffm_up_bwd_hip(*args)
torch.cuda.synchronize()
lib = _lib.load()
n = 2 * 32 * 16
buf = (ctypes.c_ulonglong * n)()
fn = lib.cabinet_debug_fz_trace if fwd else lib.cabinet_debug_xw_trace
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = fn(buf, n)
assert rc == 0, rc
names = ["store_dx", "advance+load", "(seg)", "dW 64 MFMA", "dX 64 MFMA", "hand_over", "store_chunk", "barrier", "->next top"]
if fwd:
    names = ["load issue", "y_low rows", "MFMA j=0", "epilogue 0", "MFMA j=1", "epilogue 1", "store_chunk", "barrier", "->next top"]
for wv in range(2):
    print(f"wave {4 * wv}:  chunk  " + "  ".join(f"{x:>12s}" for x in names[:8]) + "   total")
    for c in range(18):
        t = [buf[(wv * 32 + c) * 16 + i] for i in range(9)]
        if t[0] == 0:
            continue
        d = [t[i + 1] - t[i] for i in range(8)]
        nxt = buf[(wv * 32 + c + 1) * 16] if c + 1 < 32 else 0
        print(f"          {c:5d}  " + "  ".join(f"{x:12d}" for x in d) + f"   {(nxt - t[0]) if nxt else 0:6d}")
