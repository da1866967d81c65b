"""CABiNet -- host-side mirror of reference ``src/models/cabinet.py``.

Drop-in for ``from src.models.cabinet import CABiNet``: same constructor signature,
``forward`` contract ``(final_logit, high_res_logit_up)``, ``get_params()`` grouping,
sub-module names and ``state_dict`` keys.  On HIP tensors ``FeatureFusionModule.forward``
(reference cabinet.py:142-153) runs as the fused gfx950 pipeline behind
:func:`cabinet_amd.functional.ffm_fused` and the CAB attention core as one fused
kernel (see ``cab.py``); ``ConvBNReLU`` runs its BatchNorm + ReLU through the fused streaming op
(and the spatial branch's 7x7/2 stem through its own MFMA kernel).  The three dense 3x3 stride-1 convolutions of the
decoder -- ``ab.conva``, the fusion head's ``ab.b1`` over ``cat([x, feat])`` and ``conv_out.conv`` (reference
cabinet.py:59, :68 + :88-89, :160) -- run as K11's fused Winograd kernels
(:func:`cabinet_amd.functional.conv3x3`; ``CABINET_CONV3X3=0`` restores MIOpen for A/B timing); the strided and
wide 1x1 convolutions of backbone and spatial branch stay on stock PyTorch-ROCm (MIOpen), as BASELINE.json's
north_star prescribes.
"""

from __future__ import annotations

import logging
from pathlib import Path
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as _functional
from ..functional import (batched_bn_counters, bn_act, bn_relu_cls, conv1x1, conv1x1_bias_supported, conv3x3, conv3x3_bn_part,
                          conv3x3_supported, ffm_fused, ffm_fused_upsampled, stem_conv, stem_conv_supported)
from .cab import ContextAggregationBlock
from .constants import MODEL_CONFIG, MOBILENETV3_CFGS
from .mobilenetv3 import MobileNetV3

logger = logging.getLogger(__name__)


def _resize(x, size):
    return F.interpolate(x, size=size, mode="bilinear", align_corners=False)


def _is_plain_3x3(conv: nn.Conv2d) -> bool:
    """3x3, stride 1, padding 1, dilation 1, one group, no bias: what K11 (conv3x3_wino.hip) computes."""
    return (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1) and
            conv.groups == 1 and conv.bias is None and conv.padding_mode == "zeros")


def _conv3x3_bn_relu(conv: nn.Conv2d, bn: nn.BatchNorm2d, x: torch.Tensor, x1: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``relu(bn(conv(cat([x, x1], 1))))`` (``x1`` optional) on device tensors: K11's fused Winograd kernels when the layer is a plain
    3x3 with channel counts inside their coverage -- the concat is then never materialised and, in training mode, the BatchNorm's
    batch statistics come out of the convolution's epilogue (no statistics pass over its output) -- else the stock convolution;
    K7 (BatchNorm + ReLU in one streaming pass) either way."""
    c1 = 0 if x1 is None else x1.shape[1]
    if _functional.CONV3X3_ENABLED and _is_plain_3x3(conv) and conv3x3_supported(x.shape[1], c1, conv.out_channels, x.shape[2], x.shape[3]):
        part = conv3x3_bn_part(x, conv.out_channels) if bn.training else None
        return bn_act(conv3x3(x, conv.weight, x1, part), bn, "relu", conv_part=part)
    return bn_act(conv(x if x1 is None else torch.cat([x, x1], dim=1)), bn, "relu")


def _conv3x3_bn_relu_cls(conv: nn.Conv2d, bn: nn.BatchNorm2d, cls: nn.Conv2d, x: torch.Tensor,
                         x1: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``cls(relu(bn(conv(cat([x, x1], 1)))))``: K11 for the 3x3 (as :func:`_conv3x3_bn_relu`), then K12 -- BatchNorm, ReLU and the
    1x1 classifier as one streaming operator that never writes the (B, 256, H', W') activation or, in backward, its gradient
    (reference cabinet.py:88-92 and :160-172); shapes outside K12's coverage take K7 + the stock 1x1 convolution."""
    c1 = 0 if x1 is None else x1.shape[1]
    if _functional.CONV3X3_ENABLED and _is_plain_3x3(conv) and conv3x3_supported(x.shape[1], c1, conv.out_channels, x.shape[2], x.shape[3]):
        part = conv3x3_bn_part(x, conv.out_channels) if bn.training else None
        return bn_relu_cls(conv3x3(x, conv.weight, x1, part), bn, cls, conv_part=part)
    return bn_relu_cls(conv(x if x1 is None else torch.cat([x, x1], dim=1)), bn, cls)


class ConvBNReLU(nn.Module):
    """conv -> BN -> ReLU, kaiming(a=1) conv init (reference cabinet.py:19-51)."""

    def __init__(self, in_chan: int, out_chan: int, kernel_size: int = 3, stride: int = 1, padding: int = 1,
                 dilation: int = 1):
        super().__init__()
        self.conv = nn.Conv2d(in_chan, out_chan, kernel_size, stride, padding, dilation=dilation, bias=False)
        self.bn = nn.BatchNorm2d(out_chan)
        self.relu = nn.ReLU(inplace=True)
        self.init_weight()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            return self.relu(self.bn(self.conv(x)))
        if x.shape[1] == 3 and stem_conv_supported(self.conv):
            # K9: the 7x7/2 image stem without NHWC round trips; K7: BatchNorm + ReLU in one streaming pass pair
            return bn_act(stem_conv(x, self.conv), self.bn, "relu")
        return _conv3x3_bn_relu(self.conv, self.bn, x)  # K11 for conv_out's 256 -> 256 3x3 (cabinet.py:160)

    def init_weight(self) -> None:
        nn.init.kaiming_normal_(self.conv.weight, a=1)
        if self.conv.bias is not None:
            nn.init.constant_(self.conv.bias, 0)


class AttentionBranch(nn.Module):
    """conva -> CAB -> convb, plus the cat -> 3x3 -> BN -> ReLU -> 1x1 head (reference cabinet.py:54-105)."""

    def __init__(self, inplanes: int, interplanes: int, outplanes: int, num_classes: int):
        super().__init__()
        self.conva = nn.Sequential(nn.Conv2d(inplanes, interplanes, 3, padding=1, bias=False),
                                   nn.BatchNorm2d(interplanes), nn.ReLU(True))
        self.a2block = ContextAggregationBlock(interplanes, interplanes // 2)
        self.convb = nn.Conv2d(interplanes, outplanes, kernel_size=1, bias=True)
        self.b1 = nn.Conv2d(inplanes + outplanes, outplanes, 3, padding=1, bias=False)
        self.b2 = nn.BatchNorm2d(outplanes)
        self.b3 = nn.ReLU(True)
        self.b4 = nn.Conv2d(outplanes, num_classes, kernel_size=1, bias=True)
        self.init_weight()

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if x.is_cuda:
            feat = self.a2block(_conv3x3_bn_relu(self.conva[0], self.conva[1], x))
            # convb (1x1 with bias, cabinet.py:65-66): the small-grid MFMA product with the bias in its epilogue
            low_res_out = conv1x1(feat, self.convb.weight, self.convb.bias) if conv1x1_bias_supported(feat, self.convb) \
                else self.convb(feat)
            # K11 reads x and feat through two pointers: the (B, inplanes + 256, H', W') concat is never written
            # ... and K12 runs b2 -> b3 -> b4 as one operator: the ReLU output is never written
            return low_res_out, _conv3x3_bn_relu_cls(self.b1, self.b2, self.b4, x, feat)
        feat = self.a2block(self.conva(x))
        low_res_out = self.convb(feat)
        high_res_out = self.b4(self.b3(self.b2(self.b1(torch.cat([x, feat], dim=1)))))
        return low_res_out, high_res_out

    def init_weight(self) -> None:
        # walks EVERY conv below, including the CAB's zero-initialised project_out (cabinet.py:96-105)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, a=1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


class SpatialBranch(nn.Module):
    """7x7/2 -> 3x3/2 -> 3x3/2 -> 1x1: 128 channels at 1/8 resolution (reference cabinet.py:108-129)."""

    def __init__(self):
        super().__init__()
        self.conv1 = ConvBNReLU(3, 64, kernel_size=7, stride=2, padding=3)
        self.conv2 = ConvBNReLU(64, 64, kernel_size=3, stride=2, padding=1)
        self.conv3 = ConvBNReLU(64, 64, kernel_size=3, stride=2, padding=1)
        self.conv_out = ConvBNReLU(64, 128, kernel_size=1, stride=1, padding=0)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.conv_out(self.conv3(self.conv2(self.conv1(x))))


class FeatureFusionModule(nn.Module):
    """cat -> 1x1 conv -> BN -> ReLU -> channel gate (reference cabinet.py:132-153)."""

    def __init__(self, in_chan: int, out_chan: int):
        super().__init__()
        self.convblk = ConvBNReLU(in_chan, out_chan, kernel_size=1, stride=1, padding=0)
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.conv1 = nn.Conv2d(out_chan, out_chan // 4, kernel_size=1, bias=False)
        self.relu = nn.ReLU(True)
        self.conv2 = nn.Conv2d(out_chan // 4, out_chan, kernel_size=1, bias=False)
        self.sigmoid = nn.Sigmoid()

    def forward(self, fsp: torch.Tensor, fcp: torch.Tensor) -> torch.Tensor:
        if fsp.is_cuda:  # fused gfx950 pipeline (K3/K4); raises if the HIP library is unusable
            return ffm_fused(fsp, fcp, self.convblk.conv.weight, self.convblk.bn, self.conv1.weight,
                             self.conv2.weight)
        feat = self.convblk(torch.cat([fsp, fcp], dim=1))
        atten = self.sigmoid(self.conv2(self.relu(self.conv1(self.avg_pool(feat)))))
        return feat * atten + feat

    def forward_upsampled(self, fsp: torch.Tensor, low: torch.Tensor) -> torch.Tensor:
        """``self(fsp, bilinear_upsample(low, fsp.shape[2:]))`` -- the pairing CABiNet.forward uses
        (reference cabinet.py:228-230, :236).  On HIP tensors the resize is fused into the FFM kernels and
        the upsampled (B,256,H/8,W/8) tensor is never materialised; host tensors take the composite path."""
        if fsp.is_cuda:
            return ffm_fused_upsampled(fsp, low, self.convblk.conv.weight, self.convblk.bn, self.conv1.weight,
                                       self.conv2.weight)
        return self.forward(fsp, _resize(low, fsp.shape[2:]))


class CABiNetOutput(nn.Module):
    """3x3 ConvBNReLU -> 1x1 classifier (reference cabinet.py:156-172)."""

    def __init__(self, in_chan: int, mid_chan: int, n_classes: int):
        super().__init__()
        self.conv = ConvBNReLU(in_chan, mid_chan, kernel_size=3, padding=1)
        self.conv_out = nn.Conv2d(mid_chan, n_classes, kernel_size=1, bias=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.is_cuda and x.shape[1] != 3:   # K11 (3x3) + K12 (BatchNorm -> ReLU -> classifier in one pass over the 3x3's output)
            return _conv3x3_bn_relu_cls(self.conv.conv, self.conv.bn, self.conv_out, x)
        return self.conv_out(self.conv(x))


_DECODER_CHILDREN = ("ffm", "conv_out", "ab")


class CABiNet(nn.Module):
    """reference cabinet.py:175-300."""

    def __init__(self, n_classes: int, backbone_weights: Optional[Path] = None, cfgs=None, mode="large"):
        super().__init__()
        if cfgs is None and mode in MOBILENETV3_CFGS:
            cfgs = MOBILENETV3_CFGS[mode]  # convenience: the reference requires cfgs from its YAML
        self.mobile = MobileNetV3(cfgs=cfgs, mode=mode, num_classes=n_classes, weights=backbone_weights)
        config = MODEL_CONFIG.get(mode)
        if config is None:
            raise ValueError(f"Invalid mode: {mode}. Must be 'large' or 'small'")
        self.attention_planes = config["attention_planes"]
        if backbone_weights is not None:
            logger.info(f"Backbone weights loaded from {backbone_weights} via MobileNetV3")
        self.ab = AttentionBranch(self.attention_planes, 256, 256, n_classes)
        self.sb = SpatialBranch()
        self.ffm = FeatureFusionModule(128 + 256, 256)
        self.conv_out = CABiNetOutput(256, 256, n_classes)

    def forward_lowres(self, x: torch.Tensor, boundary: Optional[list] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Both heads at H/8 x W/8, i.e. ``forward`` without its two final x8 bilinear upsamples (reference
        cabinet.py:240-245).  ``OhemCELoss.forward_upsampled`` fuses exactly those resizes into the loss.
        ``boundary`` (a list) receives the two tensors that separate the decoder (``ab``, ``ffm``, ``conv_out``) from the
        encoders (``sb``, ``mobile``): the data-parallel step back-propagates the two halves separately so that the
        decoder's gradient all-reduce overlaps the encoders' backward (cabinet_amd.train.GraphedDDPStep)."""
        with batched_bn_counters():  # one multi-tensor `num_batches_tracked += 1` for the model's 59 BatchNorms
            feat_sb = self.sb(x)
            mob = self.mobile(x)
            if boundary is not None:
                boundary.extend([feat_sb, mob])
            low, high = self.ab(mob)
            high_up = _resize(high, feat_sb.shape[2:])
            final = self.conv_out(self.ffm.forward_upsampled(feat_sb, low))
        return final, high_up

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        final, high_up = self.forward_lowres(x)   # resize of `low` fused into the FFM
        size = x.shape[2:]
        return _resize(final, size), _resize(high_up, size)

    def get_params(self):
        """(wd, no_wd, lr_mul_wd, lr_mul_no_wd) with the reference's type-based rule (cabinet.py:249-300):
        Conv2d weights decay, Conv2d biases / BatchNorm / anything else (e.g. CAB.gamma) do not; the
        children ``ffm``, ``conv_out`` and ``ab`` form the x10-LR decoder groups."""
        groups = {False: ([], []), True: ([], [])}
        for name, child in self.named_children():
            decay, no_decay = groups[name in _DECODER_CHILDREN]
            seen = set()
            for m in child.modules():
                if isinstance(m, nn.Conv2d):
                    decay.append(m.weight)
                    seen.add(id(m.weight))
                    if m.bias is not None:
                        no_decay.append(m.bias)
                        seen.add(id(m.bias))
                elif isinstance(m, nn.BatchNorm2d):
                    for p in m.parameters():
                        no_decay.append(p)
                        seen.add(id(p))
            no_decay.extend(p for p in child.parameters() if id(p) not in seen)
        (wd, nowd), (lr_wd, lr_nowd) = groups[False], groups[True]
        return wd, nowd, lr_wd, lr_nowd
