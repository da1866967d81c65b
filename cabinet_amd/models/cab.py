"""Context Aggregation Block -- host-side mirror of reference ``src/models/cab.py``.

Same class names, constructor signatures, sub-module names (hence ``state_dict`` keys,
``get_params`` grouping and ``init_weight`` behaviour) as the reference, so train/eval
code written against ``src.models.cab`` drives it unchanged.  What differs is the
execution of the hot span: on HIP tensors the affinity/softmax/aggregation of
``GlobalContextAttention.forward`` (reference cab.py:149-154) runs as ONE fused gfx950
kernel through :func:`cabinet_amd.functional.cab_attention`; the n x n attention matrix
is never materialised and q/k/v are consumed in their native NCHW-flattened layout
(no transposes).  Parameters stay inside ``nn.Conv2d`` / ``nn.BatchNorm2d`` containers on
purpose -- the reference's optimizer grouping and initialisation walk those types.
"""

from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..functional import (cab_attention, cab_attention_proj, cab_local, cab_local_supported, cab_qkv, cab_qkv_supported)


def _conv_bn_relu_1x1(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class DWConv(nn.Module):
    """3x3 depthwise conv + BN + ReLU (reference cab.py:18-38)."""

    def __init__(self, channels, stride=1):
        super().__init__()
        dw = nn.Conv2d(channels, channels, kernel_size=3, stride=stride, padding=1, groups=channels, bias=False)
        self.block = nn.Sequential(dw, nn.BatchNorm2d(channels), nn.ReLU(inplace=True))

    def forward(self, x):
        return self.block(x)


class PSPModule(nn.Module):
    """Pyramid pooling with identity branch, (B,C,H,W) -> (B,C,H,W) (reference cab.py:46-76)."""

    def __init__(self, in_channels, sizes=(1, 3, 6, 8)):
        super().__init__()
        self.stages = nn.ModuleList(nn.AdaptiveAvgPool2d((s, s)) for s in sizes)
        self.project = nn.Conv2d(in_channels * (len(sizes) + 1), in_channels, kernel_size=1, bias=False)

    def forward(self, x):
        size = x.shape[2:]
        pyramid = [x]
        pyramid += [F.interpolate(stage(x), size=size, mode="bilinear", align_corners=False)
                    for stage in self.stages]
        return self.project(torch.cat(pyramid, dim=1))


class GlobalContextAttention(nn.Module):
    """Non-local attention with PSP-encoded keys/values (reference cab.py:84-162)."""

    def __init__(self, in_channels, key_channels, value_channels, out_channels=None, scale=1,
                 psp_sizes=(1, 3, 6, 8)):
        super().__init__()
        self.scale = scale
        self.out_channels = out_channels or in_channels
        self.pool = nn.MaxPool2d(kernel_size=scale) if scale > 1 else nn.Identity()
        self.to_query = _conv_bn_relu_1x1(in_channels, key_channels)
        self.to_key = _conv_bn_relu_1x1(in_channels, key_channels)
        self.to_value = nn.Conv2d(in_channels, value_channels, 1, bias=False)
        self.psp_key = PSPModule(key_channels, psp_sizes)
        self.psp_value = PSPModule(value_channels, psp_sizes)
        self.project_out = nn.Conv2d(value_channels, self.out_channels, kernel_size=1, bias=False)
        nn.init.constant_(self.project_out.weight, 0)  # reference cab.py:129

    def _native_producers(self, xd):
        wq, wv = self.to_query[0].weight, self.to_value.weight
        sizes = [st.output_size[0] for st in self.psp_key.stages]
        return (wq.shape[0] % 16 == 0 and self.project_out.weight.shape[0] % 16 == 0
                and cab_qkv_supported(xd, wq.shape[0], wv.shape[0], sizes))

    def forward(self, x):
        b, _, h, w = x.shape
        xd = self.pool(x)
        hd, wd = xd.shape[2:]
        n = hd * wd
        if xd.is_cuda and self._native_producers(xd):
            # K6: projections, BatchNorm, ReLU and both pyramid poolings as a short chain of MFMA GEMMs and
            # plane passes (no concat, no full-resolution pyramid maps), then K1 with project_out (cab.py:155) applied to
            # each context tile in its epilogue -- one launch; K1 + a small GEMM for the shapes the fused form does not take
            q, k, v = cab_qkv(xd, self)
            ctx = cab_attention_proj(q, k, v, self.project_out.weight, k.shape[1] ** -0.5).reshape(b, -1, hd, wd)
            if self.scale > 1:
                ctx = F.interpolate(ctx, size=(h, w), mode="bilinear", align_corners=False)
            return ctx
        # NCHW-flattened operands: q,k (B,Kc,n), v (B,Vc,n); the reference's transposes
        # (cab.py:138,146,154) only exist to feed torch.bmm and are not needed here
        q = self.to_query(xd).reshape(b, -1, n)
        k = self.psp_key(self.to_key(xd)).reshape(b, -1, n)
        v = self.psp_value(self.to_value(xd)).reshape(b, -1, n)
        ctx = cab_attention(q, k, v, k.shape[1] ** -0.5)  # fused K1/K2 on HIP tensors
        ctx = self.project_out(ctx.reshape(b, -1, hd, wd))
        if self.scale > 1:
            ctx = F.interpolate(ctx, size=(h, w), mode="bilinear", align_corners=False)
        return ctx


class LocalAttention(nn.Module):
    """x + x * sigmoid(DW3x3 x3 (x))  (reference cab.py:170-184)."""

    def __init__(self, channels):
        super().__init__()
        self.refine = nn.Sequential(DWConv(channels), DWConv(channels), DWConv(channels))
        self.gate = nn.Sigmoid()

    def forward(self, x):
        if x.is_cuda and cab_local_supported(x):  # one channel-resident kernel each way (K5)
            return cab_local(x, self.refine)
        return x + x * self.gate(self.refine(x))


class ContextAggregationBlock(nn.Module):
    """gamma * global(x) + local(x)  (reference cab.py:192-216)."""

    def __init__(self, in_channels, value_channels):
        super().__init__()
        self.global_attn = GlobalContextAttention(in_channels=in_channels, key_channels=in_channels // 2,
                                                  value_channels=value_channels, out_channels=in_channels,
                                                  scale=1)
        self.local_attn = LocalAttention(in_channels)
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        if x.is_cuda and cab_local_supported(x):  # local branch and the gamma-combine in the same kernel
            return cab_local(x, self.local_attn.refine, self.global_attn(x), self.gamma)
        return self.gamma * self.global_attn(x) + self.local_attn(x)
