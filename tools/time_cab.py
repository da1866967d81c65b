"""Time ContextAggregationBlock pieces (fwd+bwd) at a given shape: whole block, global branch, attention core."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cabinet_amd.functional import cab_attention
from cabinet_amd.models.cab import ContextAggregationBlock


def timeit(fn, iters=30):
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
    torch.manual_seed(0)
    m = ContextAggregationBlock(C, C // 2).cuda().train()
    torch.nn.init.kaiming_normal_(m.global_attn.project_out.weight)
    with torch.no_grad():
        m.gamma.fill_(0.5)
    x = torch.randn(B, C, H, W, device="cuda", requires_grad=True)
    g = torch.randn_like(x)
    n = H * W
    q = torch.randn(B, C // 2, n, device="cuda", requires_grad=True)
    k = torch.randn(B, C // 2, n, device="cuda", requires_grad=True)
    v = torch.randn(B, C // 2, n, device="cuda", requires_grad=True)
    gc = torch.randn(B, C // 2, n, device="cuda")

    def block():
        x.grad = None
        m(x).backward(g)

    def glob():
        x.grad = None
        m.global_attn(x).backward(g)

    def core():
        cab_attention(q, k, v, (C // 2) ** -0.5).backward(gc)

    def local():
        x.grad = None
        m.local_attn(x).backward(g)

    print(f"shape {(B, C, H, W)} fwd+bwd us: block {timeit(block):.0f}  global branch {timeit(glob):.0f}  "
          f"attention core {timeit(core):.0f}  local branch {timeit(local):.0f}")


if __name__ == "__main__":
    main()
