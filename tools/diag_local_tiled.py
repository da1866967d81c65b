"""Diagnostic: K5 tiled form vs the fp64 oracle, per channel, at 16 x 256 x 32 x 32 (training)."""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_gpu_local import _make, _oracle  # noqa: E402

B, C, H, W = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (16, 256, 32, 32)
m = _make(C, 7).cuda().train()
ref_m = _make(C, 7)
g0 = torch.Generator().manual_seed(3)
x = torch.randn(B, C, H, W, generator=g0).cuda().requires_grad_(True)
g = torch.randn(B, C, H, W, generator=g0)
out = m(x)
out.backward(g.cuda())
torch.cuda.synchronize()
o_out, o_dx, o_grads, o_buf, _ = _oracle(ref_m, x, g, True)
o32 = _oracle(ref_m, x, g, True, dtype=torch.float32)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


print("out", rel(out, o_out), "dx", rel(x.grad, o_dx))
for name, p in m.named_parameters():
    e = (p.grad.double().cpu() - o_grads[name].double()).flatten(1).norm(dim=1) if p.grad.dim() > 1 else (p.grad.double().cpu() - o_grads[name].double()).abs()
    print(f"{name:28s} gpu {rel(p.grad, o_grads[name]):.2e} cpu32 {rel(o32[2][name], o_grads[name]):.2e}  worst channels",
          [(int(i), f"{float(e[i]):.1e}") for i in e.argsort(descending=True)[:3]], "|ref| %.2e" % float(o_grads[name].norm()))
