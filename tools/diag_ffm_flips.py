"""Where the FFM backward's in-situ distance from fp64 comes from.

The real model runs its real step at a BASELINE configuration (tests/insitu.py captures what enters and leaves the fused-upsample
FFM, reference cabinet.py:142-153 + :228-230).  On exactly those tensors, on the GPU in fp64:

  z      the pre-BatchNorm product of the HIP forward against W [fsp; U(low)] in fp64 (and against a library fp32 product)
  mask   the units whose ReLU decision differs from the fp64 forward's -- with the kernel's own expression
         fma((z - mean) invstd, gamma, beta) > 0 on its own z / statistics, and with the GPU's z under fp64 statistics
  grads  the model's dfsp / dlow / parameter gradients against the fp64 replay under (i) the fp64 mask (= the in-situ row),
         (ii) the kernel's OWN mask (= the kernels' arithmetic alone)

    python tools/diag_ffm_flips.py [--config 3|5] [--out gpurun_out/ffm_flips_config3.json]
"""
import argparse
import copy
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def replay(sd, fsp, low, g, mask=None, eps=1e-5):
    """fp64 FFM forward + backward on the device; mask=None: the fp64 forward's own ReLU, else feat = pre * mask."""
    W = sd["ffm.convblk.conv.weight"].double().cuda().flatten(1).requires_grad_(True)
    gam = sd["ffm.convblk.bn.weight"].double().cuda().requires_grad_(True)
    bet = sd["ffm.convblk.bn.bias"].double().cuda().requires_grad_(True)
    w1 = sd["ffm.conv1.weight"].double().cuda().flatten(1).requires_grad_(True)
    w2 = sd["ffm.conv2.weight"].double().cuda().flatten(1).requires_grad_(True)
    fo = fsp.double().requires_grad_(True)
    lo = low.double().requires_grad_(True)
    x = torch.cat([fo, F.interpolate(lo, size=fo.shape[2:], mode="bilinear", align_corners=False)], 1)
    B, C, H, Wd = x.shape
    z = torch.matmul(W, x.flatten(2)).view(B, -1, H, Wd)
    mean = z.mean(dim=(0, 2, 3), keepdim=True)
    var = z.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    xhat = (z - mean) * (var + eps).rsqrt()
    pre = xhat * gam.view(1, -1, 1, 1) + bet.view(1, -1, 1, 1)
    m64 = pre.detach() > 0
    feat = pre * (m64 if mask is None else mask).double()
    a = feat.mean(dim=(2, 3))
    a = torch.sigmoid(F.relu(a @ w1.t()) @ w2.t())
    out = feat * a[:, :, None, None] + feat
    out.backward(g.double())
    grads = {"dfsp": fo.grad, "dlow": lo.grad, "convblk.conv.weight": W.grad, "convblk.bn.weight": gam.grad,
             "convblk.bn.bias": bet.grad, "conv1.weight": w1.grad, "conv2.weight": w2.grad}
    return dict(z=z.detach(), mean=mean.detach().flatten(), invstd=(var + eps).rsqrt().detach().flatten(), pre=pre.detach(),
                mask=m64, out=out.detach(), grads=grads)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    mode, batch, height, width, ncls = {3: ("large", 8, 1024, 1024, 8), 5: ("large", 2, 2048, 1024, 19),
                                        2: ("small", 4, 512, 512, 8), 6: ("large", 2, 512, 512, 19)}[a.config]
    from insitu import instrument, own_relu_output
    from cabinet_amd import functional as Fn
    from cabinet_amd.loss import ohem_upsampled_pair
    from cabinet_amd.train import build_model, make_criteria, synthetic_batch

    net = build_model(mode, n_classes=ncls, seed=0, gamma=0.5)
    sd = copy.deepcopy(net.state_dict())
    im, lb = synthetic_batch(batch, height, width, ncls, "cpu", seed=1)
    net = net.cuda().train()
    cap = instrument(net)
    crit_p, crit_16 = make_criteria(batch, height, width, "cuda")
    low, low16 = net.forward_lowres(im.cuda())
    loss = ohem_upsampled_pair(crit_p, low, crit_16, low16, lb.cuda(), (height, width))
    loss.backward()
    torch.cuda.synchronize()
    grads = {k[len("ffm."):]: p.grad for k, p in net.named_parameters() if k.startswith("ffm.") and p.grad is not None}
    fsp, lo, g = cap["ffm.fsp"], cap["ffm.low"], cap["d.ffm.y"]
    # the HIP forward once more on the captured tensors (bit-reproducible kernels): its z and statistics
    wb = sd["ffm.convblk.conv.weight"].cuda().flatten(1).contiguous()
    bw, bb = sd["ffm.convblk.bn.weight"].cuda(), sd["ffm.convblk.bn.bias"].cuda()
    rm, rv = sd["ffm.convblk.bn.running_mean"].cuda().clone(), sd["ffm.convblk.bn.running_var"].cuda().clone()
    out, z, mean, invstd, pooled, gate = Fn.ffm_up_fwd_hip(fsp, lo, wb, bw, bb, rm, rv, sd["ffm.conv1.weight"].cuda().flatten(1).contiguous(),
                                                           sd["ffm.conv2.weight"].cuda().flatten(1).contiguous(), True, 0.1, 1e-5,
                                                           Fn._ffm_precision())
    assert torch.equal(out, cap["ffm.y"]), "the replayed HIP forward differs from the model's"
    r64 = replay(sd, fsp, lo, g)
    res = dict(config=a.config, units=z.numel())
    res["z_gpu_vs_f64"] = rel(z, r64["z"])
    # a library fp32 product on the materialised concat (hipBLASLt / rocBLAS summation order) as the yardstick of "fp32 z"
    x32 = torch.cat([fsp, F.interpolate(lo, size=fsp.shape[2:], mode="bilinear", align_corners=False)], 1)
    z_lib = torch.matmul(wb, x32.flatten(2)).view_as(z)
    res["z_lib32_vs_f64"] = rel(z_lib, r64["z"])
    zc = z.cpu()
    torch.set_num_threads(os.cpu_count() or 8)
    z_cpu = F.conv2d(x32.cpu(), sd["ffm.convblk.conv.weight"])
    res["z_cpu32_vs_f64"] = rel(z_cpu.cuda(), r64["z"])
    res["mean_gpu_vs_f64"], res["invstd_gpu_vs_f64"] = rel(mean, r64["mean"]), rel(invstd, r64["invstd"])
    # masks
    own = own_relu_output(z, mean, invstd, bw, bb) > 0
    m64 = r64["mask"]
    res["flips_own_mask"] = int((own != m64).sum())
    # where the remaining flips sit: channel, |pre| of the fp64 forward, |mean| / sigma of the channel (the fp32 product's error is
    # relative to |z| ~ |mean| + sigma, the decision band to sigma)
    fl = (own != m64).nonzero()
    res["flipped_units"] = [dict(b=int(i[0]), c=int(i[1]), y=int(i[2]), x=int(i[3]), pre64=float(r64["pre"][tuple(i)]),
                                 z_gpu_minus_z64=float(z[tuple(i)].double() - r64["z"][tuple(i)]),
                                 mean_over_sigma=float(r64["mean"][int(i[1])] * r64["invstd"][int(i[1])])) for i in fl[:16]]
    ms = (r64["mean"] * r64["invstd"]).abs()
    res["channels_mean_over_sigma"] = dict(max=float(ms.max()), median=float(ms.median()))

    def mask_of(zz):   # this z under fp64 statistics and fp64 arithmetic: what z's error alone flips
        pre = (zz.double() - r64["mean"].view(1, -1, 1, 1)) * r64["invstd"].view(1, -1, 1, 1) * bw.double().view(1, -1, 1, 1) \
            + bb.double().view(1, -1, 1, 1)
        return pre > 0

    res["flips_from_z_gpu_alone"] = int((mask_of(z) != m64).sum())
    res["flips_from_z_lib32_alone"] = int((mask_of(z_lib) != m64).sum())
    res["flips_from_z_cpu32_alone"] = int((mask_of(z_cpu.cuda()) != m64).sum())
    # the CPU path's own BatchNorm + ReLU mask (ATen: y = z * alpha + beta' per channel)
    with torch.no_grad():
        feat_cpu = F.relu(F.batch_norm(z_cpu, None, None, sd["ffm.convblk.bn.weight"], sd["ffm.convblk.bn.bias"], True, 0.1, 1e-5))
    res["flips_cpu32_path"] = int(((feat_cpu > 0).cuda() != m64).sum())
    del z_cpu, feat_cpu, zc, x32, z_lib
    # absolute z error next to the decision boundary vs everywhere (same thing for a sequential chain; recorded)
    near = r64["pre"].abs() < 1e-3
    res["z_abs_err_rms_near_boundary"] = float((z.double() - r64["z"])[near].pow(2).mean().sqrt())
    res["z_abs_err_rms_all"] = float((z.double() - r64["z"]).pow(2).mean().sqrt())
    res["z_rms"] = float(r64["z"].pow(2).mean().sqrt())
    gpu = dict(dfsp=cap["d.ffm.fsp"], dlow=cap["d.ffm.low"], **grads)
    rows = {}
    for k, v in r64["grads"].items():
        rows[k] = dict(gpu_vs_f64_mask_replay=rel(gpu[k].flatten(), v.flatten()))
    del r64
    rown = replay(sd, fsp, lo, g, mask=own)
    for k, v in rown["grads"].items():
        rows[k]["gpu_vs_own_mask_replay"] = rel(gpu[k].flatten(), v.flatten())
    rows["out"] = dict(gpu_vs_own_mask_replay=rel(out, rown["out"]))
    res["grads"] = rows
    print(json.dumps(res, indent=1))
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
