"""K11 (conv3x3_wino.hip) against MIOpen on the same box: correctness vs an fp64 CPU convolution at a small grid, then HIP-event
times of forward / data gradient / weight gradient at the three production shapes of config 3 (and config 5 with --config5).

    python tools/time_conv3x3.py [--config5] [--iters 20]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from cabinet_amd.functional import conv3x3_bwd_hip, conv3x3_fwd_hip


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def check(B, C0, C1, K, H, W, dev):
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + W)
    x0 = torch.randn(B, C0, H, W, generator=g)
    x1 = torch.randn(B, C1, H, W, generator=g) if C1 else None
    w = torch.randn(K, C0 + C1, 3, 3, generator=g) * (2.0 / (9 * (C0 + C1))) ** 0.5
    dy = torch.randn(B, K, H, W, generator=g)
    xin = (torch.cat([x0, x1], 1) if C1 else x0).double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y_ref = F.conv2d(xin, wd, padding=1)
    y_ref.backward(dy.double())
    y = conv3x3_fwd_hip(x0.to(dev), x1.to(dev) if C1 else None, w.to(dev))
    dx0, dx1, dw = conv3x3_bwd_hip(dy.to(dev), x0.to(dev), x1.to(dev) if C1 else None, w.to(dev))
    torch.cuda.synchronize()
    e = {"y": rel(y, y_ref), "dx0": rel(dx0, xin.grad[:, :C0]), "dw": rel(dw, wd.grad)}
    if C1:
        e["dx1"] = rel(dx1, xin.grad[:, C0:])
    print(f"check B={B} C0={C0} C1={C1} K={K} {H}x{W}: " + " ".join(f"{k}={v:.2e}" for k, v in e.items()), flush=True)
    return max(e.values())


def timeit(fn, iters):
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


def bench(name, B, C0, C1, K, H, W, dev, iters):
    x0 = torch.randn(B, C0, H, W, device=dev)
    x1 = torch.randn(B, C1, H, W, device=dev) if C1 else None
    w = torch.randn(K, C0 + C1, 3, 3, device=dev) * 0.02
    dy = torch.randn(B, K, H, W, device=dev)
    xc = torch.cat([x0, x1], 1) if C1 else x0
    gf = 2.0 * B * H * W * (C0 + C1) * K * 9 / 1e9
    t = {}
    t["fwd"] = timeit(lambda: conv3x3_fwd_hip(x0, x1, w), iters)
    t["dgrad"] = timeit(lambda: conv3x3_bwd_hip(dy, x0, x1, w, need_dx=True, need_dw=False), iters)
    t["wgrad"] = timeit(lambda: conv3x3_bwd_hip(dy, x0, x1, w, need_dx=False, need_dw=True), iters)
    m = {}
    m["fwd"] = timeit(lambda: F.conv2d(xc, w, padding=1), iters)
    m["dgrad"] = timeit(lambda: torch.ops.aten.convolution_backward(dy, xc, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                                   [True, False, False]), iters)
    m["wgrad"] = timeit(lambda: torch.ops.aten.convolution_backward(dy, xc, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                                   [False, True, False]), iters)
    if C1:
        m["cat"] = timeit(lambda: torch.cat([x0, x1], 1), iters)
    for k in ("fwd", "dgrad", "wgrad"):
        print(f"{name:10s} {k:6s} K11 {t[k]:8.1f} us ({gf / t[k] * 1e3:6.1f} TFLOP/s eff, {gf / 2.25 / t[k] * 1e3 / 157.3:5.3f} of fp32 MFMA)"
              f"   MIOpen {m[k]:8.1f} us ({gf / m[k] * 1e3:6.1f} TFLOP/s eff)   ratio {m[k] / t[k]:.2f}x", flush=True)
    if C1:
        print(f"{name:10s} cat    MIOpen path also pays {m['cat']:.1f} us for torch.cat", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config5", action="store_true")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-check", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    if not a.no_check:
        worst = 0.0
        for shp in [(2, 64, 0, 64, 8, 8), (1, 64, 64, 64, 6, 34), (2, 128, 0, 64, 7, 9), (1, 64, 0, 128, 33, 20),
                    (2, 192, 64, 128, 16, 32)]:
            worst = max(worst, check(*shp, dev))
        print("worst relative error", worst, flush=True)
        assert worst < 1e-4, worst
    if a.config5:
        shapes = [("conva", 2, 960, 0, 256, 64, 32), ("b1", 2, 960, 256, 256, 64, 32), ("conv_out", 2, 256, 0, 256, 256, 128)]
    else:
        shapes = [("conva", 8, 960, 0, 256, 32, 32), ("b1", 8, 960, 256, 256, 32, 32), ("conv_out", 8, 256, 0, 256, 128, 128)]
    for s in shapes:
        bench(*s, dev, a.iters)
