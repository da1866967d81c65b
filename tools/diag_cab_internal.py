"""Diagnostic: forward values and gradients of q / k / v / ctx inside the CAB, GPU model vs fp32 and fp64 CPU oracle
(Large 2x512^2; `train` or eval-mode BN on calibrated statistics)."""
import copy
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
import cabinet_amd.functional as Fh  # noqa: E402
import cabinet_amd.models.cab as cabmod  # noqa: E402
from cabinet_amd.train import build_model, make_criteria, synthetic_batch  # noqa: E402
from oracle import model_ref  # noqa: E402

mode, batch, size, ncls = "large", 2, 512, 19
train = len(sys.argv) > 1 and sys.argv[1] == "train"
net = build_model(mode, n_classes=ncls, seed=0, gamma=0.5, freeze_unused=False)
if not train:
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 1.0
    net.train()
    with torch.no_grad():
        net(synthetic_batch(batch, size, size, ncls, "cpu", seed=7)[0])
    net.eval()
sd = copy.deepcopy(net.state_dict())
im, lb = synthetic_batch(batch, size, size, ncls, "cpu", seed=1)


def run_oracle(dt):
    cap = {}

    def ga(w, x, training, pre):
        b, _, h, wd = x.shape
        n = h * wd
        q = F.relu(model_ref._bn(w, F.conv2d(x, w[pre + "to_query.0.weight"]), pre + "to_query.1", training))
        k = F.relu(model_ref._bn(w, F.conv2d(x, w[pre + "to_key.0.weight"]), pre + "to_key.1", training))
        k = model_ref._psp(w, k, pre + "psp_key")
        v = model_ref._psp(w, F.conv2d(x, w[pre + "to_value.weight"]), pre + "psp_value")
        q, k, v = q.reshape(b, -1, n), k.reshape(b, -1, n), v.reshape(b, -1, n)
        s = torch.bmm(q.transpose(1, 2), k) * (k.shape[1] ** -0.5)
        ctx = torch.bmm(v, F.softmax(s, dim=-1).transpose(1, 2))
        for t in (q, k, v, ctx):
            t.retain_grad()
        cap.update(q=q, k=k, v=v, ctx=ctx)
        return F.conv2d(ctx.reshape(b, -1, h, wd), w[pre + "project_out.weight"])
    old = model_ref._global_attn
    model_ref._global_attn = ga
    w = model_ref.Weights(sd, dtype=dt)
    out, out16 = model_ref.cabinet_forward(w, im.to(dt), mode, training=train)
    n_min = max(1, batch * size * size // 16)
    (model_ref.ohem_ce(out, lb, 0.7, n_min) + model_ref.ohem_ce(out16, lb, 0.7, n_min)).backward()
    model_ref._global_attn = old
    return {k: (v.detach(), v.grad) for k, v in cap.items()}, w.grads()


(o64, g64), (o32, g32) = run_oracle(torch.float64), run_oracle(torch.float32)
net = net.cuda()
cap = {}
orig = Fh.cab_attention


def spy(q, k, v, scale):
    for t in (q, k, v):
        t.retain_grad()
    out = orig(q, k, v, scale)
    out.retain_grad()
    cap.update(q=q, k=k, v=v, ctx=out)
    return out


cabmod.cab_attention = spy
crit = make_criteria(batch, size, size, "cuda")
out, out16 = net(im.cuda())
(crit[0](out, lb.cuda()) + crit[1](out16, lb.cuda())).backward()
torch.cuda.synchronize()


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


for k in ("q", "k", "v", "ctx"):
    print(f"{k:4s} value gpu {rel(cap[k], o64[k][0]):.2e} cpu32 {rel(o32[k][0], o64[k][0]):.2e} | grad gpu {rel(cap[k].grad, o64[k][1]):.2e}"
          f" cpu32 {rel(o32[k][1], o64[k][1]):.2e}  |grad| {float(o64[k][1].norm()):.3e}")
rows = sorted(((rel(p.grad, g64[n]), rel(g32[n], g64[n]), n) for n, p in net.named_parameters()
               if p.grad is not None and n in g64 and float(g64[n].norm()) > 1e-12), reverse=True)
print("worst parameter gradients (gpu vs f64, cpu32 vs f64):")
for r in rows[:14]:
    print("   %.2e %.2e %s" % r)
print("tensors past 1e-3:", sum(r[0] > 1e-3 for r in rows), "of", len(rows))

# ---- where does the error of d(beta_q) = sum_p dq * 1[q > 0] come from: the mask, dq, or the reduction?
qg, q64 = cap["q"].detach().double().cpu(), o64["q"][0]
dqg, dq64 = cap["q"].grad.double().cpu(), o64["q"][1]
flips = (qg > 0) != (q64 > 0)
print("q mask flips gpu vs fp64:", int(flips.sum()), "of", flips.numel(), "; cpu32 vs fp64:", int(((o32["q"][0] > 0) != (q64 > 0)).sum()))
true_db = (dq64 * (q64 > 0)).sum(dim=(0, 2))
print("d beta_q from (gpu dq, gpu mask): %.2e ; (gpu dq, fp64 mask): %.2e ; (fp64 dq, gpu mask): %.2e ; cpu32 dq+mask: %.2e" % (
    rel((dqg * (qg > 0)).sum(dim=(0, 2)), true_db), rel((dqg * (q64 > 0)).sum(dim=(0, 2)), true_db),
    rel((dq64 * (qg > 0)).sum(dim=(0, 2)), true_db), rel((o32["q"][1].double() * (o32["q"][0] > 0)).sum(dim=(0, 2)), true_db)))
gpu_db = dict(net.named_parameters())["ab.a2block.global_attn.to_query.1.bias"].grad.double().cpu()
print("K6's d beta_q vs the same sum formed on the host from the GPU's own dq and q: %.2e ; vs fp64 %.2e" % (
    rel(gpu_db, (dqg * (qg > 0)).sum(dim=(0, 2))), rel(gpu_db, g64["ab.a2block.global_attn.to_query.1.bias"])))
e = (dqg - dq64)
print("dq error: per-(b,c) mean / rms of the error:", float(e.mean(dim=2).abs().mean()), float(e.pow(2).mean().sqrt()),
      " dq rms", float(dq64.pow(2).mean().sqrt()))
nz = (dqg != 0) & (qg <= 0)
print("gpu dq nonzero where q == 0 (K2 does not mask; autograd's dq is the gradient at q, masked later):", int(nz.sum()))
