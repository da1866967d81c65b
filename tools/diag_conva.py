"""Diagnostic (not a test): accuracy of ab.conva (MIOpen 3x3 conv + K7 BatchNorm/ReLU) forward/backward on the tensors the
Large 2x512^2 model really feeds it, vs fp64 on the CPU."""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from cabinet_amd.functional import bn_act  # noqa: E402
from cabinet_amd.train import build_model, make_criteria, synthetic_batch  # noqa: E402

mode, batch, size, ncls = "large", 2, 512, 19
net = build_model(mode, n_classes=ncls, seed=0, gamma=0.5, freeze_unused=False).cuda().train()
im, lb = synthetic_batch(batch, size, size, ncls, "cuda", seed=1)
cap = {}
ab = net.ab
h = [ab.register_forward_pre_hook(lambda m, a: cap.__setitem__("x", a[0].detach().clone())),
     ab.a2block.register_forward_pre_hook(lambda m, a: cap.__setitem__("feat_in", a[0].detach().clone())),
     ab.a2block.register_full_backward_hook(lambda m, gi, go: cap.__setitem__("du", gi[0].detach().clone()))]
crit = make_criteria(batch, size, size, "cuda")
out, out16 = net(im)
(crit[0](out, lb) + crit[1](out16, lb)).backward()
torch.cuda.synchronize()
x, du = cap["x"], cap["du"]  # mobile output (2,960,16,16); gradient at conva's BN+ReLU output
w = ab.conva[0].weight.detach()


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


res = {}
for tag, dev, dt in (("gpu", "cuda", torch.float32), ("cpu32", "cpu", torch.float32), ("cpu64", "cpu", torch.float64)):
    xx = x.detach().to(dev, dt).clone().requires_grad_(True)
    ww = w.detach().to(dev, dt).clone().requires_grad_(True)
    bn = torch.nn.BatchNorm2d(256).to(dev, dt).train()
    z = F.conv2d(xx, ww, None, 1, 1)
    z.retain_grad()
    y = bn_act(z, bn, "relu") if dev == "cuda" else F.relu(bn(z))
    y.backward(du.to(dev, dt))
    res[tag] = dict(z=z.detach(), y=y.detach(), dz=z.grad, dx=xx.grad, dw=ww.grad, dgamma=bn.weight.grad, dbeta=bn.bias.grad)
for k in res["gpu"]:
    print(f"{k:7s} gpu {rel(res['gpu'][k], res['cpu64'][k]):.2e}   cpu32 {rel(res['cpu32'][k], res['cpu64'][k]):.2e}   |.|={float(res['cpu64'][k].norm()):.3e}")
# the same BN backward fed with the fp64-exact z (isolates K7 from the convolution's forward error)
bn = torch.nn.BatchNorm2d(256).cuda().train()
z = res["cpu64"]["z"].float().cuda().requires_grad_(True)
bn_act(z, bn, "relu").backward(du)
print("K7 alone on fp64-rounded z: dz", rel(z.grad, res["cpu64"]["dz"]), "dbeta", rel(bn.bias.grad, res["cpu64"]["dbeta"]),
      "dgamma", rel(bn.weight.grad, res["cpu64"]["dgamma"]))
zz = res["cpu64"]["z"]
yy = res["cpu64"]["y"]
print("pre-activations within 1e-6 of zero:", int(((zz - zz.mean((0, 2, 3), keepdim=True)).abs() < 1e-6).sum()), "of", zz.numel())
flip = ((res["gpu"]["y"].cpu() > 0) != (yy > 0))
print("ReLU mask flips gpu vs fp64:", int(flip.sum()), " cpu32 vs fp64:", int(((res["cpu32"]["y"] > 0) != (yy > 0)).sum()))
print("|du| at flipped positions:", float(du.cpu()[flip].abs().sum()), " total |dbeta| norm", float(res["cpu64"]["dbeta"].norm()))
