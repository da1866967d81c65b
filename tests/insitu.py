"""In-situ capture for the GPU parity tests: what the REAL model feeds the hot path, and what comes back.

``instrument(net)`` wraps, on this one model instance, the three places the section-8 hot path is entered --
``ab.a2block`` (the CAB: K6 -> K1/K2 -> conv1x1 -> K5), ``ffm.forward_upsampled`` (the fused-upsample FFM) and the two
low-resolution logits that enter the fused OHEM heads -- and records, with tensor hooks, every input, output and the
gradients that the model's own backward sends through them.  ``replay_*`` then runs the fp64 (or fp32) CPU oracle ON EXACTLY
THOSE TENSORS, so that an operator's own error is separated from whatever noise its inputs already carry (ReLU-mask flips
upstream, MIOpen's convolutions): the comparison is operator-in / operator-out, in the model, at the size the model runs.

Reference spans replayed: src/models/cab.py:131-162,182-184,213-216; src/models/cabinet.py:142-153 (+ :228-230);
src/utils/loss.py:38-80 with cabinet.py:240-245.
"""
import torch
import torch.nn.functional as F

from oracle import model_ref


def instrument(net):
    cap = {}
    # one live capture per process: a registry entry of an earlier model (keyed by id(weight): ids are reused after garbage
    # collection) must not route a later model's tensors into a stale dict
    _K11_REGISTRY.clear()
    _K12_REGISTRY.clear()
    _FFM_REGISTRY.clear()

    def keep(name, t):
        cap[name] = t.detach().clone()
        if t.requires_grad:
            t.register_hook(lambda g, n=name: cap.__setitem__("d." + n, g.detach().clone()))
        return t

    cab, cab_fwd = net.ab.a2block, net.ab.a2block.forward
    ffm_up, lowres = net.ffm.forward_upsampled, net.forward_lowres

    def cab_forward(x):
        keep("cab.x", x)
        return keep("cab.y", cab_fwd(x))

    def ffm_forward_upsampled(fsp, low):
        keep("ffm.fsp", fsp)
        keep("ffm.low", low)
        return keep("ffm.y", ffm_up(fsp, low))

    def forward_lowres(x, boundary=None):
        final, high_up = lowres(x, boundary)
        return keep("head.low", final), keep("head16.low", high_up)

    cab.forward = cab_forward
    net.ffm.forward_upsampled = ffm_forward_upsampled
    net.forward_lowres = forward_lowres

    def mobile_hook(module, args, out):
        keep("mob", out)

    net.mobile.register_forward_hook(mobile_hook)
    # the fusion head next to the CAB (reference cabinet.py:88-92): b1's output and the ReLU output that feeds b4

    def b1_hook(module, args, out):   # the stock convolution (CABINET_CONV3X3=0, or channel counts outside K11's coverage)
        keep("ab.b1o", out)

    def b4_pre_hook(module, args):
        keep("ab.r", args[0])

    net.ab.b1.register_forward_hook(b1_hook)
    net.ab.b4.register_forward_pre_hook(b4_pre_hook)
    # K11 (round 5): the three plain 3x3 convolutions of the decoder run through cabinet_amd.functional.conv3x3 with the layer's
    # weight, not through nn.Conv2d.forward, so they are captured at that call: inputs, output and the gradients autograd sends back
    _K11_REGISTRY.update({id(net.ab.conva[0].weight): (cap, keep, "ab.conva"), id(net.ab.b1.weight): (cap, keep, "ab.b1"),
                          id(net.conv_out.conv.conv.weight): (cap, keep, "conv_out.conv")})
    _install_k11_spy()
    # K12 (round 5): b2 -> b3 -> b4 and conv_out's BatchNorm -> ReLU -> classifier run as ONE operator that never writes the ReLU
    # output: b4's pre-hook above does not fire.  The spy rebuilds what the hook would have seen from what the operator saved --
    # r = relu(bn(z)) with the operator's own mean / invstd and its own mask, d(r) = W^T d(logits) -- so that the tables below keep
    # their meaning (own-mask replay = the operator's arithmetic; fp64-mask replay = arithmetic + flipped units).
    _K12_REGISTRY.update({id(net.ab.b4.weight): (cap, "ab"), id(net.conv_out.conv_out.weight): (cap, "conv_out")})
    _install_k12_spy()
    # the FFM's pre-BatchNorm product and saved statistics (round 6): what its kernels decide their ReLU mask on
    _FFM_REGISTRY[id(net.ffm.convblk.conv.weight)] = cap
    _install_ffm_spy()
    return cap


_FFM_REGISTRY = {}


def _install_ffm_spy():
    import cabinet_amd.functional as Fn

    if getattr(Fn.ffm_up_fwd_hip, "_insitu_spy", False):
        return
    orig = Fn.ffm_up_fwd_hip

    def ffm_up_fwd_spy(fsp, low, w_blk, bn_w, *rest):
        res = orig(fsp, low, w_blk, bn_w, *rest)
        for cap in _FFM_REGISTRY.values():   # one live capture at a time (instrument() clears the registries)
            if cap.get("ffm.fsp") is not None and cap["ffm.fsp"].shape == fsp.shape and "ffm.z" not in cap:
                cap["ffm.z"], cap["ffm.bn_mean"], cap["ffm.bn_invstd"] = (t.detach().clone() for t in res[1:4])
        return res

    ffm_up_fwd_spy._insitu_spy = True
    Fn.ffm_up_fwd_hip = ffm_up_fwd_spy


_K11_REGISTRY = {}


def _install_k11_spy():
    import cabinet_amd.models.cabinet as cm

    if getattr(cm.conv3x3, "_insitu_spy", False):
        return
    orig = cm.conv3x3

    def conv3x3_spy(x, weight, x1=None, bn_part=None):
        hit = _K11_REGISTRY.get(id(weight))
        if hit is None:
            return orig(x, weight, x1, bn_part)
        cap, keep, name = hit
        # a VIEW of each input goes into the convolution: the hook on the view sees the gradient of THIS use only (the mobile
        # features feed conva and b1, the CAB output feeds convb and b1: a hook on the tensor itself would see the sums)
        x = keep(name + ".x", x.view_as(x))
        if x1 is not None:
            x1 = keep(name + ".x1", x1.view_as(x1))
        y = keep(name + ".y", orig(x, weight, x1, bn_part))
        if name == "ab.b1":
            cap["ab.b1o"] = cap["ab.b1.y"]
            y.register_hook(lambda g: cap.__setitem__("d.ab.b1o", g.detach().clone()))
        return y

    conv3x3_spy._insitu_spy = True
    cm.conv3x3 = conv3x3_spy


_K12_REGISTRY = {}


def own_relu_output(z, mean, invstd, gamma, beta):
    """relu(bn(z)) as K7 / K12 evaluate it -- pre = fmaf((z - mean) * invstd, gamma, beta), fp32 -- with THEIR mask: the difference and
    the product are single fp32 operations (emulated exactly by fp32 tensor ops), the fused multiply-add is emulated in fp64 (the
    product of two fp32 numbers is exact there and the sum keeps its sign), so `pre > 0` here is `pre > 0` in the kernel."""
    v = lambda t: t.detach().view(1, -1, 1, 1)
    xh = (z.detach() - v(mean)) * v(invstd)
    pre = xh.double() * v(gamma).double() + v(beta).double()
    return pre.clamp_min(0).float()


def _install_k12_spy():
    import cabinet_amd.models.cabinet as cm

    if getattr(cm.bn_relu_cls, "_insitu_spy", False):
        return
    orig = cm.bn_relu_cls

    def bn_relu_cls_spy(z, bn, cls, conv_part=None):
        y = orig(z, bn, cls, conv_part)
        hit = _K12_REGISTRY.get(id(cls.weight))
        if hit is None or y.grad_fn is None or type(y.grad_fn).__name__ != "_BnClsBackward":
            return y   # not instrumented, or the unfused path ran (then b4's own pre-hook captured the ReLU output)
        cap, name = hit
        C = z.shape[1]
        tab = y.grad_fn.saved_tensors[1].view(C, -1)
        kt = tab.shape[1] - 8
        cap[name + ".bn_mean"], cap[name + ".bn_invstd"] = tab[:, kt].detach().clone(), tab[:, kt + 1].detach().clone()
        cap[name + ".r"] = own_relu_output(z, tab[:, kt], tab[:, kt + 1], bn.weight, bn.bias)
        cap[name + ".hi"] = y.detach().clone()
        w = cls.weight.detach().double().flatten(1)

        def hook(g):
            cap["d." + name + ".hi"] = g.detach().clone()
            cap["d." + name + ".r"] = torch.einsum("kc,bkhw->bchw", w, g.detach().double()).float()

        y.register_hook(hook)
        return y

    bn_relu_cls_spy._insitu_spy = True
    cm.bn_relu_cls = bn_relu_cls_spy


def replay_cls(weight, bias, r, d_hi):
    """The 1x1 classifier behind the ReLU (reference cabinet.py:92 / :172) in fp64 on the operator's own ReLU output `r` and the captured
    gradient of its logits: -> logits, dw, dbias."""
    w = weight.detach().cpu().double().flatten(1)
    a, g = r.detach().cpu().double(), d_hi.detach().cpu().double()
    y = torch.einsum("kc,bchw->bkhw", w, a)
    if bias is not None:
        y = y + bias.detach().cpu().double().view(1, -1, 1, 1)
    return y, torch.einsum("bkhw,bchw->kc", g, a).view_as(weight), g.sum(dim=(0, 2, 3))


def replay_conv3x3(weight, x, x1, g, dtype):
    """The reference's nn.Conv2d(3x3, padding=1, bias=False) (cabinet.py:59, :88-89, :160) on captured tensors:
    -> y, dx, dx1 (None without a second input), dw."""
    xo = x.detach().cpu().to(dtype).requires_grad_(True)
    x1o = x1.detach().cpu().to(dtype).requires_grad_(True) if x1 is not None else None
    w = weight.detach().cpu().to(dtype).requires_grad_(True)
    y = torch.nn.functional.conv2d(torch.cat([xo, x1o], 1) if x1o is not None else xo, w, padding=1)
    y.backward(g.detach().cpu().to(dtype))
    return y.detach(), xo.grad, (x1o.grad if x1o is not None else None), w.grad


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    den = float(b.norm())
    return float((a - b).norm()) / den if den > 0 else float((a - b).norm())


def _sub_state(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def replay_cab(sd, x, g, dtype, training=True):
    """CAB oracle on a captured (input, output gradient): -> out, dx, {param: grad}  (keys relative to ab.a2block.)."""
    w = model_ref.Weights(_sub_state(sd, "ab.a2block."), dtype=dtype)
    xo = x.detach().cpu().to(dtype).requires_grad_(True)
    y = model_ref.cab_forward(w, xo, training)
    y.backward(g.detach().cpu().to(dtype))
    return y.detach(), xo.grad, w.grads()


def replay_ffm(sd, fsp, low, g, dtype, training=True):
    """FFM(fsp, bilinear(low)) oracle: -> out, dfsp, dlow, {param: grad}  (keys relative to ffm.)."""
    w = model_ref.Weights(_sub_state(sd, "ffm."), dtype=dtype)
    fo = fsp.detach().cpu().to(dtype).requires_grad_(True)
    lo = low.detach().cpu().to(dtype).requires_grad_(True)
    y = model_ref.ffm_forward(w, fo, model_ref._bilinear(lo, fo.shape[2:]), training)
    y.backward(g.detach().cpu().to(dtype))
    return y.detach(), fo.grad, lo.grad, w.grads()


def replay_ffm_masked(sd, fsp, low, g, mask, training=True, eps=1e-5):
    """The FFM (cabinet.py:142-153 behind the x4 resize of :228-230) in fp64 with the ReLU replaced by a GIVEN 0/1 mask (the
    kernels' own): what separates the model's gradients from this replay is the kernels' arithmetic alone, what separates this
    replay from replay_ffm's is the flipped units.  -> dfsp, dlow, {param: grad}, mask of the fp64 forward"""
    dt = torch.float64
    p = {k: sd["ffm." + k].detach().cpu().to(dt).requires_grad_(True) for k in
         ("convblk.conv.weight", "convblk.bn.weight", "convblk.bn.bias", "conv1.weight", "conv2.weight")}
    fo = fsp.detach().cpu().to(dt).requires_grad_(True)
    lo = low.detach().cpu().to(dt).requires_grad_(True)
    z = F.conv2d(torch.cat([fo, model_ref._bilinear(lo, fo.shape[2:])], 1), p["convblk.conv.weight"])
    if training:
        mean, var = z.mean(dim=(0, 2, 3), keepdim=True), z.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    else:
        mean = sd["ffm.convblk.bn.running_mean"].to(dt).view(1, -1, 1, 1)
        var = sd["ffm.convblk.bn.running_var"].to(dt).view(1, -1, 1, 1)
    pre = (z - mean) * (var + eps).rsqrt() * p["convblk.bn.weight"].view(1, -1, 1, 1) + p["convblk.bn.bias"].view(1, -1, 1, 1)
    mask64 = pre.detach() > 0
    feat = pre * mask.detach().cpu().to(dt)
    a = feat.mean(dim=(2, 3), keepdim=True)
    a = torch.sigmoid(F.conv2d(F.relu(F.conv2d(a, p["conv1.weight"])), p["conv2.weight"]))
    (feat * a + feat).backward(g.detach().cpu().to(dt))
    return fo.grad, lo.grad, {k: v.grad for k, v in p.items()}, mask64


def replay_head(low, labels, size, n_min, dtype, thresh=0.7):
    """One loss head: OHEM-CE of the x8 bilinear upsample of the low-resolution logits: -> loss, dlow."""
    lo = low.detach().cpu().to(dtype).requires_grad_(True)
    loss = model_ref.ohem_ce(model_ref._bilinear(lo, size), labels.cpu(), thresh, n_min)
    loss.backward()
    return float(loss.detach()), lo.grad


def replay_b2(sd, z, r_gpu, d_r, training=True, eps=1e-5, prefix="ab.b2"):
    """The fusion head's BatchNorm + ReLU (reference cabinet.py:90-91; here K7 ``bn_act``) on the captured b1 output ``z``
    and the captured gradient ``d_r`` of its ReLU output, in fp64, TWICE: with the mask the fp64 forward produces and with
    the mask the GPU's own forward produced (``r_gpu > 0``).  A gradient that equals the own-mask replay is exact BatchNorm /
    ReLU backward arithmetic; whatever separates it from the fp64-mask replay is flipped units, not arithmetic.
    -> dict(out, flips, units, own=(dz, dgamma, dbeta), f64=(dz, dgamma, dbeta))"""
    z = z.detach().cpu().double()
    g = d_r.detach().cpu().double()
    gamma = sd[prefix + ".weight"].double().view(1, -1, 1, 1)
    beta = sd[prefix + ".bias"].double().view(1, -1, 1, 1)
    if training:
        mean = z.mean(dim=(0, 2, 3), keepdim=True)
        var = z.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    else:
        mean = sd[prefix + ".running_mean"].double().view(1, -1, 1, 1)
        var = sd[prefix + ".running_var"].double().view(1, -1, 1, 1)
    invstd = (var + eps).rsqrt()
    xhat = (z - mean) * invstd
    y = gamma * xhat + beta
    mask64 = y > 0
    mask_own = r_gpu.detach().cpu() > 0

    def backward(mask):
        dy = g * mask
        dgamma, dbeta = (dy * xhat).sum(dim=(0, 2, 3)), dy.sum(dim=(0, 2, 3))
        if training:
            dz = gamma * invstd * (dy - dy.mean(dim=(0, 2, 3), keepdim=True) - xhat * (dy * xhat).mean(dim=(0, 2, 3), keepdim=True))
        else:
            dz = gamma * invstd * dy
        return dz, dgamma, dbeta

    return dict(out=y.clamp_min(0), flips=int((mask64 != mask_own).sum()), units=mask64.numel(),
                own=backward(mask_own), f64=backward(mask64))


B2_OWN_MASK_TOL = 1e-5  # K7's backward against the fp64 replay with K7's own ReLU mask
OWN_MASK_ROWS = ("ab.b2.dx_own_mask", "conv_out.bn.dx_own_mask", "ffm.dfsp_own_mask", "ffm.dlow_own_mask",
                 "ffm.convblk.conv.weight_own_mask", "ffm.convblk.bn.bias_own_mask")
FFM_VS_CPU32 = 1.5   # an FFM row may be at most this multiple of the fp32 CPU oracle's distance from fp64 (same tensors) ...
FP32_LEVEL = 1e-6    # ... unless it sits at the fp32 rounding level anyway


def cab_table(net, sd, cap, training=True):
    """The CAB rows of operator_table alone (fp64 replay only): cheap enough for the full-step tests, whose gradient rule
    asks which CAB gradients are exact in situ (tests/parity_rules.py)."""
    rows = {}
    grads = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    y64, dx64, g64 = replay_cab(sd, cap["cab.x"], cap["d.cab.y"], torch.float64, training)
    for name, gpu, f64 in [("cab.out", cap["cab.y"], y64), ("cab.dx", cap["d.cab.x"], dx64)] + \
            [("ab.a2block." + k, grads["ab.a2block." + k], v) for k, v in g64.items()]:
        rows[name] = dict(gpu_vs_f64=rel(gpu, f64), norm=float(f64.double().norm()), numel=f64.numel())
    return rows


def operator_table(net, sd, cap, labels, size, n_min, with_fp32=True, training=True):
    """Every output, input gradient and parameter gradient of the three hot-path entries, as the model produced them,
    against the fp64 oracle replayed on the model's own captured tensors.  rows: name -> {gpu_vs_f64[, cpu32_vs_f64], norm}"""
    rows = {}

    def put(name, gpu, f64, f32=None):
        r = dict(gpu_vs_f64=rel(gpu, f64), norm=float(f64.double().norm()), numel=f64.numel())
        if f32 is not None:
            r["cpu32_vs_f64"] = rel(f32, f64)
        rows[name] = r

    grads = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    dts = (torch.float64, torch.float32) if with_fp32 else (torch.float64,)
    # ---- CAB
    res = {dt: replay_cab(sd, cap["cab.x"], cap["d.cab.y"], dt, training) for dt in dts}
    y64, dx64, g64 = res[torch.float64]
    y32, dx32, g32 = res.get(torch.float32, (None, None, {}))
    put("cab.out", cap["cab.y"], y64, y32)
    put("cab.dx", cap["d.cab.x"], dx64, dx32)
    for k, v in g64.items():
        put("ab.a2block." + k, grads["ab.a2block." + k], v, g32.get(k))
    # ---- FFM (fused upsample)
    res = {dt: replay_ffm(sd, cap["ffm.fsp"], cap["ffm.low"], cap["d.ffm.y"], dt, training) for dt in dts}
    y64, df64, dl64, g64 = res[torch.float64]
    y32, df32, dl32, g32 = res.get(torch.float32, (None, None, None, {}))
    put("ffm.out", cap["ffm.y"], y64, y32)
    put("ffm.dfsp", cap["d.ffm.fsp"], df64, df32)
    put("ffm.dlow", cap["d.ffm.low"], dl64, dl32)
    for k, v in g64.items():
        put("ffm." + k, grads["ffm." + k], v, g32.get(k))
    del res
    if "ffm.z" in cap:
        # round 6: the same gradients against the fp64 replay under the kernels' OWN ReLU mask (= their arithmetic alone), and the
        # number of units whose decision differs from the fp64 forward's (ffm.hip re-decides borderline units in double)
        sdg = {k: (v.to(cap["ffm.z"].device) if torch.is_tensor(v) else v) for k, v in sd.items() if k.startswith("ffm.convblk.bn.")}
        own = own_relu_output(cap["ffm.z"], cap["ffm.bn_mean"], cap["ffm.bn_invstd"], sdg["ffm.convblk.bn.weight"],
                              sdg["ffm.convblk.bn.bias"]) > 0
        df, dl, gp, mask64 = replay_ffm_masked(sd, cap["ffm.fsp"], cap["ffm.low"], cap["d.ffm.y"], own, training)
        put("ffm.dfsp_own_mask", cap["d.ffm.fsp"], df)
        put("ffm.dlow_own_mask", cap["d.ffm.low"], dl)
        put("ffm.convblk.conv.weight_own_mask", grads["ffm.convblk.conv.weight"], gp["convblk.conv.weight"])
        put("ffm.convblk.bn.bias_own_mask", grads["ffm.convblk.bn.bias"], gp["convblk.bn.bias"])
        rows["ffm.dfsp_own_mask"].update(flipped_units_vs_f64_mask=int((own.cpu() != mask64).sum()), units=mask64.numel())
        del own, df, dl, gp, mask64
    # ---- K11: the decoder's three plain 3x3 convolutions (conva, the two-pointer fusion head b1, conv_out.conv), when they ran
    # through K11 (captured at the conv3x3 call, each input through a view of its own: the gradients are this use's alone)
    for name, wkey in (("ab.conva", "ab.conva.0.weight"), ("ab.b1", "ab.b1.weight"), ("conv_out.conv", "conv_out.conv.conv.weight")):
        if name + ".y" not in cap or "d." + name + ".y" not in cap:
            continue
        x1 = cap.get(name + ".x1")
        y64, dx64, dx164, dw64 = replay_conv3x3(sd[wkey], cap[name + ".x"], x1, cap["d." + name + ".y"], torch.float64)
        put(name + ".out", cap[name + ".y"], y64)
        put(name + ".dw", grads[wkey], dw64)
        put(name + ".dx", cap["d." + name + ".x"], dx64)
        if x1 is not None:
            put(name + ".dx1", cap["d." + name + ".x1"], dx164)
        del y64, dx64, dx164, dw64
    # ---- the fusion head's BatchNorm + ReLU next to the CAB (K7; outside section 8, but the chain tables put the entry point
    # of the gradient noise between d.ab.r and d.ab.b1o: own-mask replay = arithmetic, fp64-mask replay = arithmetic + flips)
    b2 = replay_b2(sd, cap["ab.b1o"], cap["ab.r"], cap["d.ab.r"], training)
    put("ab.b2.out", cap["ab.r"], b2["out"])
    put("ab.b2.dx_own_mask", cap["d.ab.b1o"], b2["own"][0])
    put("ab.b2.weight_own_mask", grads["ab.b2.weight"], b2["own"][1])
    put("ab.b2.bias_own_mask", grads["ab.b2.bias"], b2["own"][2])
    rows["ab.b2.dx_own_mask"].update(
        flipped_units_vs_f64_mask=b2["flips"], units=b2["units"], gpu_vs_f64_mask_replay=rel(cap["d.ab.b1o"], b2["f64"][0]),
        own_mask_vs_f64_mask_replay=rel(b2["own"][0], b2["f64"][0]))
    del b2
    # ---- K12 (round 5): where b2 -> b3 -> b4 and conv_out's BatchNorm -> ReLU -> classifier ran as one operator, its logits and the
    # classifier's gradients against the fp64 replay on the operator's own ReLU output, and conv_out's BatchNorm rows as b2's above
    if "ab.hi" in cap:
        y64, dw64, db64 = replay_cls(sd["ab.b4.weight"], sd.get("ab.b4.bias"), cap["ab.r"], cap["d.ab.hi"])
        put("ab.b4.out", cap["ab.hi"], y64)
        put("ab.b4.weight", grads["ab.b4.weight"], dw64)
        put("ab.b4.bias", grads["ab.b4.bias"], db64)
    if "conv_out.hi" in cap and "conv_out.conv.y" in cap:
        y64, dw64, _ = replay_cls(sd["conv_out.conv_out.weight"], None, cap["conv_out.r"], cap["d.conv_out.hi"])
        put("conv_out.conv_out.out", cap["conv_out.hi"], y64)
        put("conv_out.conv_out.weight", grads["conv_out.conv_out.weight"], dw64)
        bo = replay_b2(sd, cap["conv_out.conv.y"], cap["conv_out.r"], cap["d.conv_out.r"], training, prefix="conv_out.conv.bn")
        put("conv_out.bn.dx_own_mask", cap["d.conv_out.conv.y"], bo["own"][0])
        put("conv_out.bn.weight_own_mask", grads["conv_out.conv.bn.weight"], bo["own"][1])
        put("conv_out.bn.bias_own_mask", grads["conv_out.conv.bn.bias"], bo["own"][2])
        rows["conv_out.bn.dx_own_mask"].update(flipped_units_vs_f64_mask=bo["flips"], units=bo["units"],
                                               own_mask_vs_f64_mask_replay=rel(bo["own"][0], bo["f64"][0]))
        del bo, y64, dw64
    # ---- the two fused OHEM heads (upstream gradient of each is exactly 1: loss = head + head16)
    losses = {}
    for name in ("head", "head16"):
        l64, d64 = replay_head(cap[name + ".low"], labels, size, n_min, torch.float64)
        l32, d32 = replay_head(cap[name + ".low"], labels, size, n_min, torch.float32) if with_fp32 else (None, None)
        put(name + ".dlow", cap["d." + name + ".low"], d64, d32)
        losses[name] = (l64, l32)
    return rows, losses


# Analytically-zero gradients: gamma of refine.0 / refine.1's BatchNorm feeds a depthwise conv + batch-statistics BatchNorm,
# whose output is invariant to a per-channel scale of its input -- the true gradient is 0 and what any fp32 implementation
# computes is the rounding noise of a cancelling sum (same rule as tests/test_gpu_fullsize.py::test_cab_block_config5_grid).
SCALE_INVARIANT = ("ab.a2block.local_attn.refine.0.block.1.weight", "ab.a2block.local_attn.refine.1.block.1.weight")


def judge_operator_table(rows, tol, training=True):
    """-> rows further than `tol` from the fp64 replay (the two scale-invariant BatchNorm gains: bounded in absolute terms
    against the same-shaped gradient next to them; they are ordinary gradients when BatchNorm runs on running statistics)."""
    scale_ref = rows["ab.a2block.local_attn.refine.2.block.1.weight"]["norm"]
    bad = {}
    for k, r in rows.items():
        if training and k in SCALE_INVARIANT:
            r["analytic_zero"] = True
            if not (r["norm"] < 1e-3 * scale_ref and r["gpu_vs_f64"] * r["norm"] < 1e-3 * scale_ref):
                bad[k] = r
        elif not r["gpu_vs_f64"] <= (B2_OWN_MASK_TOL if k in OWN_MASK_ROWS else tol):
            bad[k] = r
        elif k.startswith("ffm.") and "cpu32_vs_f64" in r and not r["gpu_vs_f64"] <= max(FFM_VS_CPU32 * r["cpu32_vs_f64"], FP32_LEVEL):
            # no FFM row may creep back to a multiple of the fp32 CPU oracle's own distance from fp64 on the
            # very same tensors (rows at the fp32 rounding level -- both sides ~2e-7 -- are exempt from the ratio)
            r["ratio_to_cpu32"] = r["gpu_vs_f64"] / r["cpu32_vs_f64"]
            bad[k] = r
    return bad
