"""Diagnostic: accuracy of MIOpen's 3x3 convolution backward-data / weight-grad / forward at the attention branch's
shapes vs fp64 on the CPU (stock-library share of the step; not a hand-written kernel)."""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
import cabinet_amd  # noqa: F401,E402  (applies the package's MIOpen environment defaults)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


for (B, Ci, Co, H, W) in [(2, 1216, 256, 16, 16), (2, 960, 256, 16, 16), (8, 1216, 256, 32, 32), (2, 1216, 256, 64, 32),
                          (4, 832, 256, 16, 16), (2, 256, 256, 64, 64), (8, 256, 256, 128, 128)]:
    g0 = torch.Generator().manual_seed(0)
    x = torch.randn(B, Ci, H, W, generator=g0).relu()
    w = torch.randn(Co, Ci, 3, 3, generator=g0) * (1.0 / (Ci * 9)) ** 0.5
    g = torch.randn(B, Co, H, W, generator=g0)
    res = {}
    for tag, dev, dt in (("gpu", "cuda", torch.float32), ("cpu32", "cpu", torch.float32), ("cpu64", "cpu", torch.float64)):
        if tag != "gpu" and B * H * W > 40000:
            continue
        xx, ww = x.detach().to(dev, dt).clone().requires_grad_(True), w.detach().to(dev, dt).clone().requires_grad_(True)
        y = F.conv2d(xx, ww, None, 1, 1)
        y.backward(g.to(dev, dt))
        res[tag] = (y.detach(), xx.grad, ww.grad)
    if "cpu64" not in res:
        continue
    print((B, Ci, Co, H, W), "fwd gpu %.1e cpu %.1e | bwd-data gpu %.1e cpu %.1e | wgrad gpu %.1e cpu %.1e" % (
        rel(res["gpu"][0], res["cpu64"][0]), rel(res["cpu32"][0], res["cpu64"][0]),
        rel(res["gpu"][1], res["cpu64"][1]), rel(res["cpu32"][1], res["cpu64"][1]),
        rel(res["gpu"][2], res["cpu64"][2]), rel(res["cpu32"][2], res["cpu64"][2])))
