"""MobileNetV3 backbone -- mirror of reference ``src/models/mobilenetv3.py``.

Out of the hot path (BASELINE.json north_star keeps the backbone's convolutions on stock
PyTorch-ROCm / MIOpen); it exists so that ``CABiNet`` is a drop-in with identical
``state_dict`` keys and an identical random-initialisation stream.  Layer order inside
every ``nn.Sequential`` and the order in which sub-modules are constructed therefore
follow the reference exactly (reference mobilenetv3.py:102-198).  The one thing that is
not stock: on device tensors every BatchNorm2d (+ ReLU / HardSwish) runs through the fused
streaming kernel written for ConvBNReLU (:func:`cabinet_amd.functional.bn_act`), because it
is the same operator and costs the step 13 ms when left to the library.
"""

from __future__ import annotations

import math
from pathlib import Path

import torch
import torch.nn as nn

__all__ = ["MobileNetV3", "InvertedResidual", "SELayer", "HardSwish", "HardSigmoid",
           "mobilenetv3_large", "mobilenetv3_small"]


def _make_divisible(v, divisor, min_value=None):
    """Round a channel count to a multiple of ``divisor`` without losing >10 % (mobilenetv3.py:18-35)."""
    floor = divisor if min_value is None else min_value
    rounded = max(floor, int(v + divisor / 2) // divisor * divisor)
    return rounded + divisor if rounded < 0.9 * v else rounded


class HardSigmoid(nn.Module):
    """relu6(x + 3) / 6  (mobilenetv3.py:38-50)."""

    def __init__(self, inplace: bool = True) -> None:
        super().__init__()
        self.relu = nn.ReLU6(inplace=inplace)

    def forward(self, x):
        return self.relu(x + 3) / 6


class HardSwish(nn.Module):
    """x * hard_sigmoid(x)  (mobilenetv3.py:53-65)."""

    def __init__(self, inplace: bool = True) -> None:
        super().__init__()
        self.sigmoid = HardSigmoid(inplace=inplace)

    def forward(self, x):
        return x * self.sigmoid(x)


class SELayer(nn.Module):
    """Squeeze-excite with hard-sigmoid gate (mobilenetv3.py:68-83)."""

    def __init__(self, channel, reduction=4):
        super().__init__()
        squeezed = _make_divisible(channel // reduction, 8)
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(nn.Linear(channel, squeezed), nn.ReLU(inplace=True),
                                nn.Linear(squeezed, channel), HardSigmoid())

    def gate(self, x):
        n, c = x.shape[:2]
        return self.fc(self.avg_pool(x).view(n, c))

    def forward(self, x):
        n, c = x.shape[:2]
        return x * self.gate(x).view(n, c, 1, 1)


def _act(use_hs):
    return HardSwish() if use_hs else nn.ReLU(inplace=True)


class _FusedSequential(nn.Sequential):
    """nn.Sequential (same children, same state_dict keys) that, on device tensors, runs every
    BatchNorm2d -> ReLU / HardSwish pair -- and every lone BatchNorm2d -- through the fused HIP op, and the
    depthwise convolutions through the HIP stencil kernels."""

    def forward(self, x, residual=None):
        if not x.is_cuda:
            y = super().forward(x)
            return y if residual is None else residual + y
        from ..functional import (bn_act, bn_act_dwconv, dwconv, dwconv_supported, gate_act, pwconv,
                                  pwconv_supported)

        layers = [m for m in self if not isinstance(m, nn.Identity)]  # placeholders of the non-SE blocks: no-ops

        def act_of(m):
            return "relu" if isinstance(m, nn.ReLU) else "hardswish" if isinstance(m, HardSwish) else None

        i = 0
        while i < len(layers):
            m = layers[i]
            nxt = layers[i + 1] if i + 1 < len(layers) else None
            if isinstance(m, nn.BatchNorm2d):
                act = act_of(nxt)
                nxt2 = layers[i + 2] if act and i + 2 < len(layers) else None
                if isinstance(nxt2, nn.Conv2d) and nxt2.groups > 1 and dwconv_supported(nxt2):
                    # BN + activation folded into the depthwise convolution's loader: the activated map never exists
                    x, i = bn_act_dwconv(x, m, act, nxt2), i + 3
                else:
                    last = i + (2 if act else 1) >= len(layers)  # the block's closing BatchNorm takes the shortcut
                    x, i = bn_act(x, m, act, residual if last else None), i + (2 if act else 1)
                    if last:
                        residual = None
            elif isinstance(m, SELayer):  # channel gate fused with the activation behind it
                act = act_of(nxt)
                x, i = gate_act(x, m.gate(x), act), i + (2 if act else 1)
            elif isinstance(m, nn.Conv2d) and m.groups == 1 and pwconv_supported(m, x):
                x, i = pwconv(x, m), i + 1  # thin 1x1 conv on a large plane: streaming MFMA kernel, no NHWC copies
            elif isinstance(m, nn.Conv2d) and m.groups > 1 and dwconv_supported(m):
                x, i = dwconv(x, m), i + 1  # depthwise stencil kernel (MIOpen only has its naive solver here)
            else:
                x, i = m(x), i + 1
        return x if residual is None else residual + x


def conv_3x3_bn(inp, oup, stride):
    return _FusedSequential(nn.Conv2d(inp, oup, 3, stride, 1, bias=False), nn.BatchNorm2d(oup), HardSwish())


def conv_1x1_bn(inp, oup):
    return _FusedSequential(nn.Conv2d(inp, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup), HardSwish())


class InvertedResidual(nn.Module):
    """MBConv block (mobilenetv3.py:102-159)."""

    def __init__(self, inp, hidden_dim, oup, kernel_size, stride, use_se, use_hs):
        super().__init__()
        if stride not in (1, 2):
            raise ValueError(f"stride must be 1 or 2, got {stride}")
        self.identity = stride == 1 and inp == oup
        pad = (kernel_size - 1) // 2

        def depthwise():
            return nn.Conv2d(hidden_dim, hidden_dim, kernel_size, stride, pad, groups=hidden_dim, bias=False)

        layers = []
        if inp == hidden_dim:  # no expansion: dw, bn, act, se, pw-linear, bn
            layers += [depthwise(), nn.BatchNorm2d(hidden_dim), _act(use_hs)]
            layers += [SELayer(hidden_dim) if use_se else nn.Identity()]
        else:  # pw, bn, act, dw, bn, se, act, pw-linear, bn
            layers += [nn.Conv2d(inp, hidden_dim, 1, 1, 0, bias=False), nn.BatchNorm2d(hidden_dim), _act(use_hs)]
            layers += [depthwise(), nn.BatchNorm2d(hidden_dim)]
            layers += [SELayer(hidden_dim) if use_se else nn.Identity(), _act(use_hs)]
        layers += [nn.Conv2d(hidden_dim, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup)]
        self.conv = _FusedSequential(*layers)

    def forward(self, x):
        if self.identity:
            return self.conv(x, residual=x)  # x + conv(x); on device the add rides on the closing BatchNorm's pass
        return self.conv(x)


class MobileNetV3(nn.Module):
    """features -> 1x1 conv; the classifier is built (for key parity) but unused (mobilenetv3.py:162-205)."""

    def __init__(self, cfgs, mode, num_classes=1000, width_mult=1.0, weights=None):
        super().__init__()
        self.cfgs = cfgs
        self.weights = weights
        if mode not in ("large", "small"):
            raise ValueError(f"mode must be 'large' or 'small', got '{mode}'")
        cin = _make_divisible(16 * width_mult, 8)
        stages = [conv_3x3_bn(3, cin, 2)]
        exp = cin
        for k, t, c, use_se, use_hs, s in self.cfgs:
            cout = _make_divisible(c * width_mult, 8)
            exp = _make_divisible(cin * t, 8)
            stages.append(InvertedResidual(cin, exp, cout, k, s, use_se, use_hs))
            cin = cout
        self.features = nn.Sequential(*stages)
        self.conv = conv_1x1_bn(cin, exp)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        head = {"large": 1280, "small": 1024}[mode]
        if width_mult > 1.0:
            head = _make_divisible(head * width_mult, 8)
        self.classifier = nn.Sequential(nn.Linear(exp, head), HardSwish(), nn.Dropout(0.2),
                                        nn.Linear(head, num_classes))
        self._initialize_weights()

    def forward(self, x):
        return self.conv(self.features(x))

    def _initialize_weights(self):
        if self.weights is not None and Path(self.weights).is_file():
            try:
                loaded = torch.load(self.weights, map_location="cpu", weights_only=True)
                merged = self.state_dict()
                merged.update({k: v for k, v in loaded.items() if "classifier" not in k})
                self.load_state_dict(merged)
                print(f"Loaded pretrained weights from {self.weights}")
                return
            except Exception as e:  # same policy as the reference: warn and fall through to random init
                print(f"Failed to load backbone weights from {self.weights}: {e}")
                print("Proceeding with random weight initialization.")
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / fan))
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, nn.Linear):
                m.weight.data.normal_(0, 0.01)
                m.bias.data.zero_()


def mobilenetv3_large(**kwargs):
    from .constants import MOBILENETV3_CFGS

    return MobileNetV3(MOBILENETV3_CFGS["large"], mode="large", **kwargs)


def mobilenetv3_small(**kwargs):
    from .constants import MOBILENETV3_CFGS

    return MobileNetV3(MOBILENETV3_CFGS["small"], mode="small", **kwargs)
