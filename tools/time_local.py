"""Time the K5 CAB-local kernels against the composite ATen chain they replace (config 3 shape by default)."""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cabinet_amd.models.cab import LocalAttention


def timeit(fn, iters=50):
    for _ in range(5):
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
    B, C, H, W = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (8, 256, 32, 32)
    m = LocalAttention(C).cuda().train()
    x = torch.randn(B, C, H, W, device="cuda", requires_grad=True)
    g = torch.randn_like(x)

    def fused():
        x.grad = None
        m(x).backward(g)

    def composite():
        x.grad = None
        (x + x * m.gate(m.refine(x))).backward(g)

    def fused_fwd():
        with torch.no_grad():
            m(x)

    def composite_fwd():
        with torch.no_grad():
            x + x * m.gate(m.refine(x))

    print(f"shape {(B, C, H, W)}  fwd+bwd: fused {timeit(fused):.1f} us  composite {timeit(composite):.1f} us   "
          f"fwd only: fused {timeit(fused_fwd):.1f} us  composite {timeit(composite_fwd):.1f} us")


if __name__ == "__main__":
    main()
