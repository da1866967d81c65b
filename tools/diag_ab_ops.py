"""Diagnostic: op-by-op backward accuracy inside AttentionBranch on the GPU (Large 2x512^2, train mode): every op's input
gradient is recomputed in fp64 on the CPU from the GPU's OWN output gradient and forward operands."""
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
net.ab.register_forward_pre_hook(lambda m, a: cap.__setitem__("x", a[0].detach().clone()))
net.ab.register_full_backward_hook(lambda m, gi, go: cap.update(dmob=gi[0].detach().clone(), dlow=go[0].detach().clone(), dhigh=go[1].detach().clone()))
crit = make_criteria(batch, size, size, "cuda")
out, out16 = net(im)
(crit[0](out, lb) + crit[1](out16, lb)).backward()
torch.cuda.synchronize()
ab = net.ab
for p in ab.parameters():
    p.grad = None
for m in ab.modules():  # replay on the same statistics state is irrelevant in train mode (batch statistics)
    pass
hk = ab.a2block.register_full_backward_hook(lambda m, gi, go: cap.update(hook_feat_in=gi[0].detach().clone(), hook_feat_out=go[0].detach().clone()))
x = cap["x"].clone().requires_grad_(True)
z = ab.conva[0](x); z.retain_grad()
y = bn_act(z, ab.conva[1], "relu"); y.retain_grad()
feat = ab.a2block(y); feat.retain_grad()
low = ab.convb(feat)
cat = torch.cat([x, feat], 1); cat.retain_grad()
b1o = ab.b1(cat); b1o.retain_grad()
r = bn_act(b1o, ab.b2, "relu"); r.retain_grad()
h = ab.b4(r)
torch.autograd.backward([low, h], [cap["dlow"], cap["dhigh"]])
torch.cuda.synchronize()


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def d(t):
    return t.detach().double().cpu()


print("replayed d mob vs the model's own:", rel(x.grad, cap["dmob"]))
print("a2block backward-hook grad_input[0] vs y.grad:", rel(cap["hook_feat_in"], y.grad), " grad_output vs feat.grad:", rel(cap["hook_feat_out"], feat.grad))
# b4: dr = W4^T dh
print("b4 bwd-data          ", rel(r.grad, F.conv_transpose2d(d(cap["dhigh"]), d(ab.b4.weight))))
# b2 (K7, train): fp64 BN+ReLU backward from the GPU's dr and b1o
b1o64 = d(b1o).requires_grad_(True)
F.relu(F.batch_norm(b1o64, None, None, d(ab.b2.weight), d(ab.b2.bias), True, 0.1, 1e-5)).backward(d(r.grad))
print("b2 K7 bwd dx         ", rel(b1o.grad, b1o64.grad))
# b1 conv bwd-data from the GPU's d b1o
cat64 = d(cat).requires_grad_(True)
F.conv2d(cat64, d(ab.b1.weight), None, 1, 1).backward(d(b1o.grad))
print("b1 bwd-data (all)    ", rel(cat.grad, cat64.grad), " x-slice", rel(cat.grad[:, :960], cat64.grad[:, :960]), " feat-slice",
      rel(cat.grad[:, 960:], cat64.grad[:, 960:]))
# conva BN (K7) and conv bwd-data
z64 = d(z).requires_grad_(True)
F.relu(F.batch_norm(z64, None, None, d(ab.conva[1].weight), d(ab.conva[1].bias), True, 0.1, 1e-5)).backward(d(y.grad))
print("conva K7 bwd dx      ", rel(z.grad, z64.grad))
x64 = d(x).requires_grad_(True)
F.conv2d(x64, d(ab.conva[0].weight), None, 1, 1).backward(d(z.grad))
tot = x64.grad + cat64.grad[:, :960]
print("conva bwd-data + b1 x-slice (fp64 from GPU upstream) vs GPU d mob:", rel(x.grad, tot))
print("   conva bwd-data alone: |fp64| %.3e ; b1 x-slice |fp64| %.3e ; sum %.3e" % (float(x64.grad.norm()), float(cat64.grad[:, :960].norm()), float(tot.norm())))
# isolate: GPU conva bwd-data alone
xa = cap["x"].clone().requires_grad_(True)
ab.conva[0](xa).backward(z.grad)
print("conva bwd-data alone (GPU vs fp64):", rel(xa.grad, x64.grad))
xb = torch.cat([cap["x"], feat.detach()], 1).requires_grad_(True)
ab.b1(xb).backward(b1o.grad)
print("b1 bwd-data alone on a fresh cat (GPU vs fp64): x-slice", rel(xb.grad[:, :960], cat64.grad[:, :960]))

# ---- forward activations of the same ops: GPU vs fp64 recomputation from the GPU's own mobile output
xg = d(cap["x"])
z_64 = F.conv2d(xg, d(ab.conva[0].weight), None, 1, 1)
y_64 = F.relu(F.batch_norm(z_64, None, None, d(ab.conva[1].weight), d(ab.conva[1].bias), True, 0.1, 1e-5))
print("forward: z", rel(z, z_64), " y", rel(y, y_64), " mask flips", int(((d(y) > 0) != (y_64 > 0)).sum()))
var = z_64.var(dim=(0, 2, 3), unbiased=False)
print("conva z per-channel var: min %.3e median %.3e max %.3e ; |mean|/std max %.2f" % (
    float(var.min()), float(var.median()), float(var.max()), float((z_64.mean(dim=(0, 2, 3)).abs() / var.sqrt()).max())))
# the oracle's own forward from the fp64 model would need the whole backbone; compare the GPU mobile output with a CPU fp32/fp64 backbone
from oracle import model_ref  # noqa: E402
import copy  # noqa: E402
net2 = build_model(mode, n_classes=ncls, seed=0, gamma=0.5, freeze_unused=False)
sd = copy.deepcopy(net2.state_dict())
for dt in (torch.float32, torch.float64):
    w = model_ref.Weights(sd, requires_grad=False, dtype=dt)
    mob = model_ref._mobilenet(w, im.cpu().to(dt), mode, True)
    print("mobile output GPU vs CPU", dt, rel(cap["x"], mob), " |mob|", float(mob.norm()))
    if dt == torch.float64:
        e = (d(cap["x"]) - mob)
        pc = e.pow(2).sum(dim=(0, 2, 3)).sqrt() / mob.pow(2).sum(dim=(0, 2, 3)).sqrt().clamp_min(1e-30)
        print("   per-channel relative error of the GPU mobile output: max %.2e median %.2e ; worst channels" % (float(pc.max()), float(pc.median())),
              [int(i) for i in pc.sort(descending=True).indices[:8]])
