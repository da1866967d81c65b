#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Run in the build container only (needs /root/reference, which never travels to
the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is produced (all fp32, torch CPU, 8 threads, seeds recorded in each file):

g1_attn_*.npz   attention core of ``GlobalContextAttention.forward``
                (reference cab.py:136-155) driven with chosen q/k/v by replacing
                the projection sub-modules with channel slices / identities, so
                the reference's own view/transpose/bmm/softmax/bmm lines run.
g2_cab.npz      ``ContextAggregationBlock(256,128)`` train + eval, gamma=0.5 and
                kaiming-initialised convs (as inside CABiNet), fwd + all grads +
                BN buffers after the train step.
g3_ffm.npz      ``FeatureFusionModule(384,256)`` train + eval, fwd + all grads +
                BN buffers.
kat_model.json  known-answer scalars of the full CABiNet (Small and Large):
                eval forward at stock init, and a train step (fwd + 2x OHEM-CE +
                bwd) with gamma=0.5: loss, logits statistics, per-parameter
                gradient norms, BN buffer statistics.

Only data is written: inputs, expected outputs, seeds.  No reference source text.
"""

import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))

import yaml  # noqa: E402
from src.models.cab import ContextAggregationBlock, GlobalContextAttention  # noqa: E402
from src.models.cabinet import CABiNet, FeatureFusionModule  # noqa: E402
from src.utils.loss import OhemCELoss  # noqa: E402

torch.set_num_threads(8)


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"wrote {name}: {os.path.getsize(path) / 1e6:.2f} MB")


class _Slice(torch.nn.Module):
    def __init__(self, lo, hi):
        super().__init__()
        self.lo, self.hi = lo, hi

    def forward(self, x):
        return x[:, self.lo:self.hi]


def gen_attn_core(tag, seed, b, kc, vc, h, w, spike=False):
    """Drive reference cab.py:136-155 with explicit q, k, v."""
    torch.manual_seed(seed)
    q = torch.randn(b, kc, h, w).relu()
    k = torch.randn(b, kc, h, w)
    v = torch.randn(b, vc, h, w)
    if spike:  # force large running-max jumps late in the key sweep (online softmax rescale)
        k[:, :, h - 1, w - 3:] *= 6.0
        q[:, :, 0, :4] *= 3.0
    g = torch.randn(b, vc, h, w)
    gca = GlobalContextAttention(kc + kc + vc, kc, vc, out_channels=vc)
    gca.to_query = _Slice(0, kc)
    gca.to_key = _Slice(kc, 2 * kc)
    gca.to_value = _Slice(2 * kc, 2 * kc + vc)
    gca.psp_key = torch.nn.Identity()
    gca.psp_value = torch.nn.Identity()
    gca.project_out = torch.nn.Identity()
    x = torch.cat([q, k, v], 1).requires_grad_(True)
    ctx = gca(x)
    ctx.backward(g)
    dq, dk, dv = x.grad[:, :kc], x.grad[:, kc:2 * kc], x.grad[:, 2 * kc:]
    npz(f"g1_attn_{tag}.npz", seed=seed, q=q, k=k, v=v, g=g, ctx=ctx, dq=dq, dk=dk, dv=dv,
        scale=np.float32(kc ** -0.5))


def _grads(mod):
    return {f"grad.{k}": p.grad for k, p in mod.named_parameters()}


def _state(mod, prefix, buffers_only=False):
    return {f"{prefix}.{k}": v.detach().clone() for k, v in mod.state_dict().items()
            if not buffers_only or k.endswith(("running_mean", "running_var", "num_batches_tracked"))}


def gen_cab():
    torch.manual_seed(10)
    cab = ContextAggregationBlock(256, 128)
    # what AttentionBranch.init_weight (reference cabinet.py:96-105) does to it inside CABiNet
    for m in cab.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight, a=1)
        elif isinstance(m, torch.nn.BatchNorm2d):
            torch.nn.init.constant_(m.weight, 1)
            torch.nn.init.constant_(m.bias, 0)
    with torch.no_grad():
        cab.gamma.fill_(0.5)
        # non-trivial BN affine + running stats so eval mode is exercised
        for m in cab.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.5, 1.5)
    torch.manual_seed(11)
    x = torch.randn(2, 256, 16, 12)
    g = torch.randn(2, 256, 16, 12)
    out = {}
    out.update(_state(cab, "init"))
    # eval first (does not touch buffers)
    cab.eval()
    xe = x.clone().requires_grad_(True)
    ye = cab(xe)
    ye.backward(g)
    out["eval.out"] = ye
    out["eval.dx"] = xe.grad
    out.update({f"eval.{k}": v.clone() for k, v in _grads(cab).items()})
    cab.zero_grad()
    cab.train()
    xt = x.clone().requires_grad_(True)
    yt = cab(xt)
    yt.backward(g)
    out["train.out"] = yt
    out["train.dx"] = xt.grad
    out.update({f"train.{k}": v.clone() for k, v in _grads(cab).items()})
    out.update(_state(cab, "after", buffers_only=True))
    npz("g2_cab.npz", x=x, g=g, **out)


def gen_ffm():
    torch.manual_seed(20)
    ffm = FeatureFusionModule(384, 256)
    with torch.no_grad():
        bn = ffm.convblk.bn
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.3, 0.3)
        bn.running_mean.uniform_(-0.2, 0.2)
        bn.running_var.uniform_(0.5, 1.5)
    torch.manual_seed(21)
    fsp = torch.randn(2, 128, 18, 15)
    fcp = torch.randn(2, 256, 18, 15)
    g = torch.randn(2, 256, 18, 15)
    out = {}
    out.update(_state(ffm, "init"))
    for mode in ("eval", "train"):
        ffm.train(mode == "train")
        ffm.zero_grad()
        a = fsp.clone().requires_grad_(True)
        c = fcp.clone().requires_grad_(True)
        y = ffm(a, c)
        y.backward(g)
        out[f"{mode}.out"] = y
        out[f"{mode}.dfsp"] = a.grad
        out[f"{mode}.dfcp"] = c.grad
        out.update({f"{mode}.{k}": v.clone() for k, v in _grads(ffm).items()})
    out.update(_state(ffm, "after", buffers_only=True))
    npz("g3_ffm.npz", fsp=fsp, fcp=fcp, g=g, **out)


def gen_model_kats():
    kats = {}
    for mode, size in (("small", 256), ("large", 256)):
        with open(f"{REF}/configs/model/mobilenetv3_{mode}.yaml") as f:
            cfgs = yaml.safe_load(f)["cfgs"]
        torch.manual_seed(0)
        net = CABiNet(n_classes=8, cfgs=cfgs, mode=mode)
        sd = net.state_dict()
        entry = {
            "model_seed": 0,
            "n_params": int(sum(p.numel() for p in net.parameters())),
            "state_keys": len(sd),
            # init fingerprint: per-tensor sum and abs-sum of a handful of tensors
            "init_fingerprint": {k: [float(sd[k].double().sum()), float(sd[k].double().abs().sum())]
                                 for k in ("mobile.features.0.0.weight", "mobile.conv.0.weight",
                                           "ab.conva.0.weight", "ab.a2block.global_attn.project_out.weight",
                                           "ab.a2block.global_attn.psp_key.project.weight",
                                           "ab.b1.weight", "sb.conv1.conv.weight",
                                           "ffm.convblk.conv.weight", "ffm.conv2.weight",
                                           "conv_out.conv_out.weight")},
        }
        # KAT-eval: stock init, eval mode
        net.eval()
        torch.manual_seed(1)
        x = torch.randn(1, 3, size, size)
        with torch.no_grad():
            out, out16 = net(x)
        entry["eval"] = {
            "data_seed": 1, "shape": [1, 3, size, size],
            "out_sum": float(out.double().sum()), "out_abs_mean": float(out.abs().mean()),
            "out_0_c_0_0": [float(t) for t in out[0, :, 0, 0]],
            "out16_sum": float(out16.double().sum()), "out16_abs_mean": float(out16.abs().mean()),
        }
        # KAT-train: gamma = 0.5 so the attention kernel matters; fwd + 2x OHEM + bwd
        with torch.no_grad():
            net.ab.a2block.gamma.fill_(0.5)
        net.train()
        torch.manual_seed(2)
        x = torch.randn(2, 3, size, size)
        lb = torch.randint(0, 8, (2, size, size))
        n_min = 2 * size * size // 16
        crit_p, crit_16 = OhemCELoss(0.7, n_min, 255), OhemCELoss(0.7, n_min, 255)
        out, out16 = net(x)
        loss = crit_p(out, lb) + crit_16(out16, lb)
        loss.backward()
        gn = {k: float(p.grad.double().norm()) for k, p in net.named_parameters() if p.grad is not None}
        entry["train"] = {
            "data_seed": 2, "shape": [2, 3, size, size], "gamma": 0.5, "n_min": n_min,
            "loss": float(loss.detach()),
            "out_sum": float(out.double().sum()), "out_abs_mean": float(out.abs().mean()),
            "out16_sum": float(out16.double().sum()), "out16_abs_mean": float(out16.abs().mean()),
            "out_slice": [float(t) for t in out[1, :, 100, 37]],
            "global_grad_norm": float(torch.sqrt(sum(p.grad.double().pow(2).sum()
                                                     for p in net.parameters() if p.grad is not None))),
            "grad_norms": gn,
            "params_without_grad": sorted(k for k, p in net.named_parameters() if p.grad is None),
            "bn_after": {k: [float(v.double().sum()), float(v.double().abs().sum())]
                         for k, v in net.state_dict().items()
                         if k.startswith(("ffm.convblk.bn.running", "ab.a2block.global_attn.to_query.1.running",
                                          "ab.a2block.local_attn.refine.2.block.1.running"))},
        }
        kats[mode] = entry
        print(mode, "loss", entry["train"]["loss"], "gnorm", entry["train"]["global_grad_norm"])
    kats["torch_version"] = torch.__version__
    with open(os.path.join(HERE, "kat_model.json"), "w") as f:
        json.dump(kats, f, indent=1, sort_keys=True)
    print("wrote kat_model.json")


if __name__ == "__main__":
    gen_attn_core("b2_k128_v128_n128", 3, 2, 128, 128, 8, 16)
    gen_attn_core("b1_k128_v128_n500_ragged", 4, 1, 128, 128, 25, 20, spike=True)
    gen_attn_core("b2_k128_v128_n64", 5, 2, 128, 128, 8, 8)
    gen_attn_core("b1_k256_v128_n96", 6, 1, 256, 128, 8, 12)
    gen_attn_core("b1_k64_v64_n77", 7, 1, 64, 64, 7, 11)
    gen_cab()
    gen_ffm()
    gen_model_kats()
