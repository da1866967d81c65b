"""Training-step harness: the reference's ``train_step`` (src/scripts/train.py:429-441)
around the MI355X-native model, single GPU or data-parallel over one node.

The reference script itself cannot run here (Hydra / torchvision are absent and its
Python never travels to the GPU box), so the step recipe is restated:

* ``n_min = max(1, B*H*W // 16)`` per process, two ``OhemCELoss(0.7, n_min, 255)``
  (train.py:329-349), ``loss = crit_p(out, lb) + crit_16(out16, lb)``, ``loss.backward()``;
* fp32 end to end.  (The reference wraps the step in autocast; the parity contract of
  BASELINE.json is against the fp32 CPU forward/backward, so autocast is off by default.)
* under torchrun: gradients are averaged with :class:`cabinet_amd.ddp.BucketedGradReducer`;
* gradient accumulation follows train.py:435-439,478-480: ``loss / accum_steps`` per micro-step, optimizer (and,
  data-parallel, the collectives) only on the last micro-step of a window, ``flush()`` for a trailing partial window.
"""

from __future__ import annotations

import contextlib

import torch

from .loss import OhemCELoss, ohem_upsampled_pair
from .models.cabinet import CABiNet
from .models.constants import DEFAULT_IGNORE_LABEL, DEFAULT_SCORE_THRESHOLD, MOBILENETV3_CFGS, OHEM_DIVISOR


def build_model(mode="large", n_classes=8, device="cpu", seed=0, gamma=None, freeze_unused=True):
    """Random-init CABiNet (model seed as in BASELINE.md); ``gamma`` overrides CAB's zero-init scale so
    the attention kernels influence logits and gradients (parity trap 1)."""
    torch.manual_seed(seed)
    net = CABiNet(n_classes=n_classes, cfgs=MOBILENETV3_CFGS[mode], mode=mode)
    if gamma is not None:
        with torch.no_grad():
            net.ab.a2block.gamma.fill_(float(gamma))
    if freeze_unused:
        # mobile.classifier never runs in forward (reference mobilenetv3.py:202-205): no grads, so
        # keep it out of the gradient buckets (DDP pitfall 1)
        for p in net.mobile.classifier.parameters():
            p.requires_grad_(False)
    return net.to(device)


def make_criteria(batch, height, width, device, thresh=DEFAULT_SCORE_THRESHOLD, ignore=DEFAULT_IGNORE_LABEL):
    n_min = max(1, batch * height * width // OHEM_DIVISOR)
    return (OhemCELoss(thresh, n_min, ignore).to(device), OhemCELoss(thresh, n_min, ignore).to(device))


class TrainStep:
    """fwd + 2x OHEM-CE + bwd (+ gradient all-reduce when a reducer is given), with the reference's accumulation
    contract: call it once per micro-batch; every ``accum_steps``-th call reduces and steps the optimizer."""

    def __init__(self, net, criteria, reducer=None, optimizer=None, autocast=False, fused_loss=None, accum_steps=1):
        self.net, (self.crit_p, self.crit_16) = net, criteria
        self.reducer, self.optimizer, self.autocast = reducer, optimizer, autocast
        # fused_loss: run the two final x8 upsamples inside the OHEM-CE kernels (device tensors only);
        # default = on whenever the model lives on a GPU
        self.fused_loss = fused_loss
        self.accum_steps = max(1, int(accum_steps))
        self._micro = 0  # micro-steps taken inside the current accumulation window

    def zero_grad(self):
        if self.reducer is not None:
            self.reducer.zero_grad()
        else:
            for p in self.net.parameters():
                p.grad = None

    def _optimizer_step(self):
        """train.py:_optimizer_step -- here: join the collectives, then step.  Gradients are zeroed at the start of the
        next window (the reference zeroes right after the step; same values, and callers can still read .grad)."""
        if self.reducer is not None:
            self.reducer.finish()
        if self.optimizer is not None:
            self.optimizer.step()
        self._micro = 0

    def flush(self):
        """Trailing partial accumulation window at the end of an epoch (train.py:478-480)."""
        if self._micro:
            self._optimizer_step()

    def __call__(self, im, lb):
        if self._micro == 0:
            self.zero_grad()
        last = self._micro + 1 == self.accum_steps
        fused = im.is_cuda if self.fused_loss is None else self.fused_loss
        sync = contextlib.nullcontext() if (last or self.reducer is None) else self.reducer.no_sync()
        with sync:
            with torch.amp.autocast(device_type=im.device.type, enabled=self.autocast):
                if fused:
                    low, low16 = self.net.forward_lowres(im)
                    size = im.shape[2:]
                    loss = ohem_upsampled_pair(self.crit_p, low, self.crit_16, low16, lb, size)
                else:
                    out, out16 = self.net(im)
                    loss = self.crit_p(out, lb) + self.crit_16(out16, lb)
                if self.accum_steps > 1:
                    loss = loss / self.accum_steps
            loss.backward()
        self._micro += 1
        if last:
            self._optimizer_step()
        return loss.detach()


def synthetic_batch(batch, height, width, n_classes, device, seed=1):
    """Inputs of BASELINE.md section 2: images ~ N(0,1), labels uniform over classes, no ignore pixels."""
    g = torch.Generator().manual_seed(seed)
    im = torch.randn(batch, 3, height, width, generator=g)
    lb = torch.randint(0, n_classes, (batch, height, width), generator=g)
    return im.to(device), lb.to(device)
