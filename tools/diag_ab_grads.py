"""Diagnostic (not a test): activation gradients around the attention branch, GPU model vs fp32/fp64 CPU oracle
(Large, 2x3x512x512, 19 classes): overall relative error and error of the per-channel sums."""
import copy
import sys

import torch

sys.path.insert(0, ".")
from cabinet_amd.train import build_model, make_criteria, synthetic_batch  # noqa: E402
from oracle import model_ref  # noqa: E402

mode, batch, size, ncls = sys.argv[1] if len(sys.argv) > 1 else "large", int(sys.argv[2]) if len(sys.argv) > 2 else 2, \
    int(sys.argv[3]) if len(sys.argv) > 3 else 512, 19
net = build_model(mode, n_classes=ncls, seed=0, gamma=0.5, freeze_unused=False)
sd = copy.deepcopy(net.state_dict())
im, lb = synthetic_batch(batch, size, size, ncls, "cpu", seed=1)

# ---- oracle with captured intermediates
orig_ab, orig_cab, orig_ffm = model_ref.attention_branch_forward, model_ref.cab_forward, model_ref.ffm_forward


VAL = {}


def run_oracle(dt):
    cap = {}

    def ab(w, x, training, pre=""):
        x.retain_grad()
        cap["mob"] = x
        low, high = orig_ab(w, x, training, pre)
        low.retain_grad(); high.retain_grad()
        cap["low"], cap["high"] = low, high
        return low, high

    def cab(w, x, training, pre=""):
        x.retain_grad()
        cap["feat_in"] = x
        y = orig_cab(w, x, training, pre)
        y.retain_grad()
        cap["feat_out"] = y
        return y

    def ffm(w, fsp, fcp, training, pre=""):
        fsp.retain_grad()
        cap["sb"] = fsp
        y = orig_ffm(w, fsp, fcp, training, pre)
        y.retain_grad()
        cap["ffm_out"] = y
        return y
    model_ref.attention_branch_forward, model_ref.cab_forward, model_ref.ffm_forward = ab, cab, ffm
    w = model_ref.Weights(sd, dtype=dt)
    model_ref.train_step(w, im.to(dt), lb, mode)
    VAL[dt] = {k: v.detach() for k, v in cap.items()}
    return {k: v.grad for k, v in cap.items()}


g64, g32 = run_oracle(torch.float64), run_oracle(torch.float32)
model_ref.attention_branch_forward, model_ref.cab_forward, model_ref.ffm_forward = orig_ab, orig_cab, orig_ffm

# ---- GPU model with hooks
net = net.cuda().train()
cap = {}


def grab(name):
    def hook(m, gi, go):
        cap[name + "_out"] = go[0].detach()
        if gi[0] is not None:
            cap[name + "_in"] = gi[0].detach()
    return hook


net.ab.register_full_backward_hook(lambda m, gi, go: cap.update(mob=gi[0].detach(), low=go[0].detach(), high=go[1].detach()))
net.ab.a2block.register_full_backward_hook(lambda m, gi, go: cap.update(feat_in=gi[0].detach(), feat_out=go[0].detach()))
net.ffm.register_full_backward_hook(lambda m, gi, go: cap.update(ffm_out=go[0].detach()))
net.sb.register_full_backward_hook(lambda m, gi, go: cap.update(sb=go[0].detach()))
crit = make_criteria(batch, size, size, "cuda")
out, out16 = net(im.cuda().requires_grad_(True))
(crit[0](out, lb.cuda()) + crit[1](out16, lb.cuda())).backward()
torch.cuda.synchronize()


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def chan(a, b):
    a, b = a.double().cpu().sum(dim=(0, 2, 3)), b.double().cpu().sum(dim=(0, 2, 3))
    return float((a - b).norm() / b.norm().clamp_min(1e-300)), float(b.norm())


for k in ("sb", "low", "high", "feat_out", "feat_in", "mob"):
    cg, nb = chan(cap[k], g64[k])
    cc, _ = chan(g32[k], g64[k])
    tot = float(g64[k].double().abs().sum())
    print(f"d {k:9s} gpu {rel(cap[k], g64[k]):.2e} cpu32 {rel(g32[k], g64[k]):.2e} | per-channel sums: gpu {cg:.2e} cpu32 {cc:.2e}"
          f"  (|sum_c| {nb:.2e} vs sum|.| {tot:.2e})")

# ---- structure of the error of d feat_in (the CAB's input gradient) and how the conva backward maps it
d_gpu, d_64, d_32 = cap["feat_in"].double().cpu(), g64["feat_in"].double(), g32["feat_in"].double()
for tag, d in (("gpu", d_gpu), ("cpu32", d_32)):
    e = d - d_64
    per_c = e.pow(2).sum(dim=(0, 2, 3)).sqrt()
    top = per_c.sort(descending=True).values
    mean_part = e.mean(dim=(0, 2, 3), keepdim=True).expand_as(e)
    print(f"[{tag}] |e|/|d| {float(e.norm() / d_64.norm()):.2e}; energy in the 8 worst channels {float(top[:8].pow(2).sum() / top.pow(2).sum()):.2f};"
          f" in the per-channel mean {float(mean_part.norm() ** 2 / e.norm() ** 2):.3f}; per-image split {[round(float(e[i].norm() ** 2 / e.norm() ** 2), 2) for i in range(e.shape[0])]}")
    sig = d_64.pow(2).sum(dim=(0, 2, 3)).sqrt()
    ratio = (per_c / sig.clamp_min(1e-30)).sort(descending=True)
    print("      worst per-channel relative errors:", [f"{float(v):.1e}@{int(i)}" for v, i in zip(ratio.values[:6], ratio.indices[:6])],
          " median", f"{float(ratio.values[len(ratio.values) // 2]):.1e}")

# ---- is d mob an ill-conditioned function of (d feat_in, d high)?  Recompute it in fp64 from the GPU's own upstream gradients.
import torch.nn.functional as F  # noqa: E402

w = {k: v.double() for k, v in sd.items()}
v64 = VAL[torch.float64]


def dmob_from(du, dhigh):
    x = v64["mob"].clone().requires_grad_(True)
    z = F.conv2d(x, w["ab.conva.0.weight"], None, 1, 1)
    y = F.relu(F.batch_norm(z, None, None, w["ab.conva.1.weight"], w["ab.conva.1.bias"], True, 0.1, 1e-5))
    b1 = F.conv2d(torch.cat([x, v64["feat_out"]], 1), w["ab.b1.weight"], None, 1, 1)
    h = F.conv2d(F.relu(F.batch_norm(b1, None, None, w["ab.b2.weight"], w["ab.b2.bias"], True, 0.1, 1e-5)), w["ab.b4.weight"], w["ab.b4.bias"])
    torch.autograd.backward([y, h], [du.double().cpu(), dhigh.double().cpu()])
    return x.grad


ref = dmob_from(g64["feat_in"], g64["high"])
print("fp64 recomputation reproduces the oracle's d mob:", rel(ref, g64["mob"]))
a = dmob_from(cap["feat_in"], cap["high"])
print("d mob recomputed in fp64 from the GPU's (d feat_in, d high): vs oracle %.2e ; vs the GPU's own d mob %.2e" % (rel(a, g64["mob"]), rel(cap["mob"], a)))
b = dmob_from(g32["feat_in"], g32["high"])
print("same from the CPU-fp32 upstream gradients: vs oracle %.2e ; vs cpu32's own d mob %.2e" % (rel(b, g64["mob"]), rel(g32["mob"], b)))
c = dmob_from(cap["feat_in"], g64["high"])
print("GPU d feat_in + exact d high: %.2e ;  exact d feat_in + GPU d high: %.2e" % (rel(c, g64["mob"]), rel(dmob_from(g64["feat_in"], cap["high"]), g64["mob"])))
