"""Per-layer timing of the model's bias-free 1x1 convolutions (fwd+bwd): stock MIOpen vs cabinet_conv1x1."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from cabinet_amd.functional import conv1x1

LAYERS = [  # (Cin, Cout, H=W) at 8x3x1024x1024, MobileNetV3-Large pointwise convs + sb.conv_out
    (16, 16, 512), (16, 64, 512), (64, 24, 256), (24, 72, 256), (72, 24, 256), (72, 40, 128), (40, 120, 128),
    (120, 40, 128), (40, 240, 128), (240, 80, 64), (80, 200, 64), (200, 80, 64), (80, 184, 64), (184, 80, 64),
    (80, 480, 64), (480, 112, 64), (112, 672, 64), (672, 112, 64), (672, 160, 32), (160, 960, 32), (960, 160, 32),
    (64, 128, 128),
]


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


def main():
    B = 8
    tot = [0.0, 0.0]
    for ci, co, h in LAYERS:
        x = torch.randn(B, ci, h, h, device="cuda", requires_grad=True)
        w = torch.randn(co, ci, 1, 1, device="cuda", requires_grad=True)
        g = torch.randn(B, co, h, h, device="cuda")

        def stock():
            x.grad = w.grad = None
            F.conv2d(x, w).backward(g)

        def ours():
            x.grad = w.grad = None
            conv1x1(x, w).backward(g)

        a, b = timeit(stock), timeit(ours)
        tot[0] += a
        tot[1] += b
        gb = 4.0 * B * h * h * (2 * ci + 2 * co + ci + co) / 1e9  # fwd r/w + bwd (dy, x read; dx write) approx
        print(f"{ci:4d}->{co:4d} @{h:3d}  stock {a:8.1f} us   conv1x1 {b:8.1f} us   ({gb / (b * 1e-6) / 1e3:.2f} TB/s eff)")
    print(f"total stock {tot[0] / 1e3:.2f} ms   conv1x1 {tot[1] / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
