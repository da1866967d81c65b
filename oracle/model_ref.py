"""Functional CPU restatement of the whole CABiNet step.  TEST INFRASTRUCTURE.

A ``state_dict`` in the reference's key layout goes in; logits come out.  No
``nn.Module`` is built: every layer is an ``F.*`` call on tensors looked up by
their reference key, so this file shares no structure with either the reference
modules or ``cabinet_amd.models`` and is a genuinely independent check.

Follows
-------
* ``/root/reference/src/models/cabinet.py:75-94,108-129,142-172,207-247``
* ``/root/reference/src/models/cab.py:65-76,131-162,182-184,213-216``
* ``/root/reference/src/models/mobilenetv3.py:44-159,202-205``
* ``/root/reference/configs/model/mobilenetv3_{large,small}.yaml`` (cfgs tables)
* ``/root/reference/src/utils/loss.py:38-80`` (OHEM-CE)
* ``/root/reference/src/scripts/train.py:329-349,429-441`` (step recipe, run in
  plain fp32: the reference's autocast would be bf16 on CPU)
"""

from __future__ import annotations

import time

import torch
import torch.nn.functional as F

# configs/model/mobilenetv3_large.yaml:5-21 / mobilenetv3_small.yaml:5-17 : k, t, c, SE, HS, s
MNV3_CFGS = {
    "large": [
        [3, 1, 16, 0, 0, 1], [3, 4, 24, 0, 0, 2], [3, 3, 24, 0, 0, 1],
        [5, 3, 40, 1, 0, 2], [5, 3, 40, 1, 0, 1], [5, 3, 40, 1, 0, 1],
        [3, 6, 80, 0, 1, 2], [3, 2.5, 80, 0, 1, 1], [3, 2.3, 80, 0, 1, 1],
        [3, 2.3, 80, 0, 1, 1], [3, 6, 112, 1, 1, 1], [3, 6, 112, 1, 1, 1],
        [5, 6, 160, 1, 1, 2], [5, 6, 160, 1, 1, 1], [5, 6, 160, 1, 1, 1],
    ],
    "small": [
        [3, 1, 16, 1, 0, 2], [3, 4.5, 24, 0, 0, 2], [3, 3.67, 24, 0, 0, 1],
        [5, 4, 40, 1, 1, 2], [5, 6, 40, 1, 1, 1], [5, 6, 40, 1, 1, 1],
        [5, 3, 48, 1, 1, 1], [5, 3, 48, 1, 1, 1], [5, 6, 96, 1, 1, 2],
        [5, 6, 96, 1, 1, 1], [5, 6, 96, 1, 1, 1],
    ],
}


def _round_channels(v, divisor=8):
    """mobilenetv3.py:18-35."""
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


class Weights:
    """Leaf tensors keyed like the reference state_dict; float params get grad."""

    def __init__(self, state_dict, requires_grad=True, dtype=torch.float32):
        self.t = {}
        for k, v in state_dict.items():
            v = v.detach().cpu().clone()
            if v.is_floating_point():
                v = v.to(dtype)
            is_buffer = k.endswith(("running_mean", "running_var", "num_batches_tracked"))
            if v.is_floating_point() and not is_buffer and requires_grad:
                v.requires_grad_(True)
            self.t[k] = v

    def __getitem__(self, k):
        return self.t[k]

    def grads(self):
        return {k: v.grad for k, v in self.t.items() if v.requires_grad and v.grad is not None}

    def buffers(self):
        return {k: v for k, v in self.t.items()
                if k.endswith(("running_mean", "running_var", "num_batches_tracked"))}


# -- primitive layers ---------------------------------------------------------


def _bn(w, x, key, training):
    y = F.batch_norm(x, w[key + ".running_mean"], w[key + ".running_var"],
                     w[key + ".weight"], w[key + ".bias"], training, 0.1, 1e-5)
    if training:
        w.t[key + ".num_batches_tracked"] += 1
    return y


def _hsig(x):
    return F.relu6(x + 3) / 6  # mobilenetv3.py:48-50


def _hswish(x):
    return x * _hsig(x)  # mobilenetv3.py:63-65


def _bilinear(x, size):
    return F.interpolate(x, size=size, mode="bilinear", align_corners=False)


# -- MobileNetV3 (mobilenetv3.py:102-205) -------------------------------------


def _mobilenet(w, x, mode, training, pre="mobile."):
    cfgs = MNV3_CFGS[mode]
    x = _hswish(_bn(w, F.conv2d(x, w[pre + "features.0.0.weight"], None, 2, 1),
                    pre + "features.0.1", training))
    cin = 16
    for idx, (k, t, c, use_se, use_hs, s) in enumerate(cfgs, start=1):
        cout = _round_channels(c)
        hid = _round_channels(cin * t)
        act = _hswish if use_hs else F.relu
        p = f"{pre}features.{idx}.conv."

        def se(v, key):
            y = v.mean(dim=(2, 3))
            y = F.relu(F.linear(y, w[key + ".fc.0.weight"], w[key + ".fc.0.bias"]))
            y = _hsig(F.linear(y, w[key + ".fc.2.weight"], w[key + ".fc.2.bias"]))
            return v * y[:, :, None, None]

        inp = x
        if cin == hid:  # dw -> bn -> act -> se -> pw-linear -> bn   (mobilenetv3.py:110-128)
            y = F.conv2d(x, w[p + "0.weight"], None, s, (k - 1) // 2, 1, hid)
            y = act(_bn(w, y, p + "1", training))
            if use_se:
                y = se(y, p + "3")
            y = _bn(w, F.conv2d(y, w[p + "4.weight"]), p + "5", training)
        else:  # pw -> bn -> act -> dw -> bn -> se -> act -> pw-linear -> bn (mobilenetv3.py:130-153)
            y = act(_bn(w, F.conv2d(x, w[p + "0.weight"]), p + "1", training))
            y = F.conv2d(y, w[p + "3.weight"], None, s, (k - 1) // 2, 1, hid)
            y = _bn(w, y, p + "4", training)
            if use_se:
                y = se(y, p + "5")
            y = act(y)
            y = _bn(w, F.conv2d(y, w[p + "7.weight"]), p + "8", training)
        x = inp + y if (s == 1 and cin == cout) else y
        cin = cout
    # mobilenetv3.py:186,202-205: features then the 1x1 "conv" block; classifier unused
    return _hswish(_bn(w, F.conv2d(x, w[pre + "conv.0.weight"]), pre + "conv.1", training))


# -- CAB (cab.py) ---------------------------------------------------------------


def _psp(w, x, key, sizes=(1, 3, 6, 8)):
    h, wd = x.shape[2:]
    priors = [x] + [_bilinear(F.adaptive_avg_pool2d(x, (s, s)), (h, wd)) for s in sizes]
    return F.conv2d(torch.cat(priors, 1), w[key + ".project.weight"])


def _global_attn(w, x, training, pre):
    b, _, h, wd = x.shape
    n = h * wd
    q = F.relu(_bn(w, F.conv2d(x, w[pre + "to_query.0.weight"]), pre + "to_query.1", training))
    k = F.relu(_bn(w, F.conv2d(x, w[pre + "to_key.0.weight"]), pre + "to_key.1", training))
    k = _psp(w, k, pre + "psp_key")
    v = _psp(w, F.conv2d(x, w[pre + "to_value.weight"]), pre + "psp_value")
    q, k, v = q.reshape(b, -1, n), k.reshape(b, -1, n), v.reshape(b, -1, n)
    s = torch.bmm(q.transpose(1, 2), k) * (k.shape[1] ** -0.5)  # cab.py:149-150
    p = F.softmax(s, dim=-1)  # cab.py:151
    ctx = torch.bmm(v, p.transpose(1, 2)).reshape(b, -1, h, wd)  # cab.py:153-154
    return F.conv2d(ctx, w[pre + "project_out.weight"])  # cab.py:155


def _local_attn(w, x, training, pre):
    y = x
    for i in range(3):  # cab.py:175-179
        c = y.shape[1]
        y = F.conv2d(y, w[f"{pre}refine.{i}.block.0.weight"], None, 1, 1, 1, c)
        y = F.relu(_bn(w, y, f"{pre}refine.{i}.block.1", training))
    return x + x * torch.sigmoid(y)  # cab.py:182-184


def cab_forward(w, x, training, pre=""):
    """ContextAggregationBlock.forward, cab.py:213-216."""
    return w[pre + "gamma"] * _global_attn(w, x, training, pre + "global_attn.") + \
        _local_attn(w, x, training, pre + "local_attn.")


# -- FFM, branches, heads (cabinet.py) -----------------------------------------


def ffm_forward(w, fsp, fcp, training, pre=""):
    """FeatureFusionModule.forward, cabinet.py:142-153."""
    z = F.conv2d(torch.cat([fsp, fcp], 1), w[pre + "convblk.conv.weight"])
    feat = F.relu(_bn(w, z, pre + "convblk.bn", training))
    a = feat.mean(dim=(2, 3), keepdim=True)
    a = F.relu(F.conv2d(a, w[pre + "conv1.weight"]))
    a = torch.sigmoid(F.conv2d(a, w[pre + "conv2.weight"]))
    return feat * a + feat


def _conv_bn_relu(w, x, key, training, stride, pad):
    return F.relu(_bn(w, F.conv2d(x, w[key + ".conv.weight"], None, stride, pad),
                      key + ".bn", training))


def _tap(taps, name, t):
    """Record an intermediate (and keep its gradient) for the in-situ parity tables; no-op without a dict."""
    if taps is not None:
        if t.requires_grad:
            t.retain_grad()
        taps[name] = t
    return t


def attention_branch_forward(w, x, training, pre="", taps=None):
    """AttentionBranch.forward, cabinet.py:75-94."""
    feat = F.relu(_bn(w, F.conv2d(x, w[pre + "conva.0.weight"], None, 1, 1),
                      pre + "conva.1", training))
    _tap(taps, "cab.x", feat)
    feat = _tap(taps, "cab.y", cab_forward(w, feat, training, pre + "a2block."))
    low = F.conv2d(feat, w[pre + "convb.weight"], w[pre + "convb.bias"])
    fused = _tap(taps, "ab.b1o", F.conv2d(torch.cat([x, feat], 1), w[pre + "b1.weight"], None, 1, 1))
    fused = _tap(taps, "ab.r", F.relu(_bn(w, fused, pre + "b2", training)))
    high = F.conv2d(fused, w[pre + "b4.weight"], w[pre + "b4.bias"])
    return low, high


def cabinet_forward(w, x, mode, training, taps=None):
    """CABiNet.forward, cabinet.py:207-247.  Returns (final_logit, high_res_logit_up).
    ``taps`` (a dict) receives the tensors either side of the CAB, the FFM and the two loss heads, with their gradients
    retained -- the quantities the in-situ parity tests compare with what the HIP model saw at the same places."""
    hh, ww = x.shape[2:]
    sb = _conv_bn_relu(w, x, "sb.conv1", training, 2, 3)  # cabinet.py:111-114,126-129
    sb = _conv_bn_relu(w, sb, "sb.conv2", training, 2, 1)
    sb = _conv_bn_relu(w, sb, "sb.conv3", training, 2, 1)
    sb = _conv_bn_relu(w, sb, "sb.conv_out", training, 1, 0)
    _tap(taps, "ffm.fsp", sb)
    mob = _tap(taps, "mob", _mobilenet(w, x, mode, training))
    low, high = attention_branch_forward(w, mob, training, "ab.", taps)
    _tap(taps, "ffm.low", low)
    low_up = _bilinear(low, sb.shape[2:])
    high_up = _tap(taps, "head16.low", _bilinear(high, sb.shape[2:]))
    fuse = _tap(taps, "ffm.y", ffm_forward(w, sb, low_up, training, "ffm."))
    final = _conv_bn_relu(w, fuse, "conv_out.conv", training, 1, 1)
    final = _tap(taps, "head.low", F.conv2d(final, w["conv_out.conv_out.weight"]))
    return _bilinear(final, (hh, ww)), _bilinear(high_up, (hh, ww))


# -- OHEM-CE and the train step ---------------------------------------------------


def ohem_ce(logits, labels, thresh, n_min, ignore_lb=255):
    """OhemCELoss.forward, loss.py:38-80 (weight=None)."""
    loss = F.cross_entropy(logits, labels, ignore_index=ignore_lb, reduction="none")
    valid = loss[labels != ignore_lb]
    if valid.numel() == 0:
        return torch.zeros((), requires_grad=True)
    srt, _ = torch.sort(valid, descending=True)
    n_min = min(int(n_min), srt.numel())
    picked = srt[srt > thresh] if srt[n_min - 1] > thresh else srt[:n_min]
    return picked.mean()


def train_step(w, x, labels, mode, thresh=0.7, ohem_divisor=16, ignore_lb=255, taps=None):
    """fwd + 2x OHEM-CE + bwd in fp32 (train.py:329-349,429-441 without autocast)."""
    out, out16 = cabinet_forward(w, x, mode, training=True, taps=taps)
    b, _, hh, ww = x.shape
    n_min = max(1, b * hh * ww // ohem_divisor)
    loss = ohem_ce(out, labels, thresh, n_min, ignore_lb) + \
        ohem_ce(out16, labels, thresh, n_min, ignore_lb)
    loss.backward()
    return out.detach(), out16.detach(), loss.detach()


def usable_cpu_threads(cap=64):
    """Host threads this process may really use: affinity mask and cgroup CPU quota, capped.

    (os.cpu_count() on the GPU box reports every hardware thread of the node although the job's
    cgroup grants a fraction of them; oversubscribing intra-op threads made the baseline ~40x slower.)
    """
    import os

    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def baseline_sample(batch, size, n_classes):
    """The bounded sample the CPU baseline is timed on (and the HIP model is checked on, in the same bench run)."""
    h, w = (size, size) if isinstance(size, int) else size
    g = torch.Generator().manual_seed(1)
    x = torch.randn(batch, 3, h, w, generator=g)
    lb = torch.randint(0, n_classes, (batch, h, w), generator=g)
    return x, lb


def time_cpu_baseline(state_dict, mode, batch, size, n_classes, steps=8, warmup=1, threads=None,
                      min_seconds=10.0, max_seconds=40.0):
    """Images/s of the CPU restatement on synthetic data (bench.py cpu_baseline leg)."""
    threads = threads or usable_cpu_threads()
    torch.set_num_threads(threads)
    x, lb = baseline_sample(batch, size, n_classes)
    times, it, loss = [], 0, None
    # bounded sample: `warmup` untimed steps, then timed steps until ~min_seconds of CPU work or `steps`
    while True:
        w = Weights(state_dict)
        t0 = time.perf_counter()
        loss = train_step(w, x, lb, mode)[2]
        dt = time.perf_counter() - t0
        if it >= warmup:
            times.append(dt)
        it += 1
        if len(times) >= steps or (len(times) >= 2 and sum(times) >= min_seconds) or \
                (times and sum(times) >= max_seconds):
            break
    dt = sum(times) / len(times)
    return dict(value=batch / dt, seconds_per_step=dt, cores=threads, batch=batch, timed_steps=len(times),
                seconds=sum(times), loss=float(loss))
