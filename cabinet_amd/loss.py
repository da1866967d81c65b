"""OHEM cross-entropy -- mirror of reference ``src/utils/loss.py:11-83``.

Not a hand-written kernel (SURVEY.md section 8(f) row f3 lists it as "next"); it is part of
the timed step (BASELINE config 3), so the semantics are restated exactly: per-pixel CE,
drop ignored pixels, keep everything above ``thresh`` if the ``n_min``-th hardest is above
it, else the ``n_min`` hardest; mean.

The reference finds the ``n_min``-th hardest value with a full descending sort of every valid
pixel (8.4 M elements per head at config 3) plus two boolean gathers.  The selection rule only
needs a COUNT, because  sorted[n_min-1] > thresh  <=>  #(loss > thresh) >= n_min :
  * count >= n_min : mean of the losses above thresh = masked sum / count   (no sort, no gather)
  * otherwise      : mean of the n_min largest valid losses = topk          (rare: late training)
One host read (two integers) replaces the reference's two data-dependent syncs.  Values and
gradients are identical up to fp32 summation order (checked against the sort-based oracle).
"""

from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class OhemCELoss(nn.Module):
    def __init__(self, thresh, n_min, ignore_lb=255, weight=None):
        super().__init__()
        self.thresh = float(thresh)
        self.n_min = int(n_min)
        self.ignore_lb = ignore_lb
        if weight is not None and not isinstance(weight, torch.Tensor):
            weight = torch.tensor(weight, dtype=torch.float32)
        self.register_buffer("weight", weight)

    def forward(self, logits, labels):
        w = self.weight if isinstance(self.weight, torch.Tensor) else None
        loss = F.cross_entropy(logits, labels, weight=w, ignore_index=self.ignore_lb, reduction="none")
        valid = labels != self.ignore_lb
        above = (loss > self.thresh) & valid
        n_valid, n_above = torch.stack([valid.sum(), above.sum()]).tolist()  # the step's one host sync
        if n_valid == 0:
            return torch.zeros((), device=logits.device, requires_grad=True)
        n_min = min(self.n_min, n_valid)
        if n_above >= n_min:  # the n_min-th hardest pixel is above thresh: keep every pixel above it
            return (loss * above).sum() / n_above
        hardest = torch.topk(loss.masked_fill(~valid, float("-inf")).flatten(), n_min, sorted=False).values
        return hardest.mean()

    def forward_upsampled(self, logits_low, labels, size=None):
        """``self(F.interpolate(logits_low, size, mode="bilinear", align_corners=False), labels)`` -- the pairing
        of reference cabinet.py:240-245 with loss.py:38-80 -- with the upsample, the softmax and the selection fused
        into hand-written kernels on HIP tensors (``cabinet_ohem_up_fwd/bwd``): neither the (B,C,H,W) logits nor
        their log-softmax nor the per-pixel gradient are materialised.  The kernels implement the branch
        "at least n_min pixels above thresh"; class weights or the top-n_min branch take the composite path."""
        prep = self._fused_launch(logits_low, labels, size)
        return self._fused_finish(prep, None)

    # The fused head runs in two phases so that a caller with several heads can issue all forward kernels first
    # and pay ONE host read-back for the branch decisions (see ohem_upsampled_pair).
    def _fused_launch(self, logits_low, labels, size):
        size = tuple(size) if size is not None else tuple(labels.shape[-2:])
        fused = logits_low.is_cuda and not isinstance(self.weight, torch.Tensor) and logits_low.shape[1] <= 32
        if not fused:
            return (logits_low, labels, size, None, None, None)
        from .functional import _f32c, ohem_up_fwd_hip

        # the kernels read `const long long*` labels of exactly (B,H,W): anything else is the caller's error, as it is
        # for F.cross_entropy ("expected scalar type Long"), never a silent reinterpretation of the bytes
        if labels.dtype != torch.int64:
            raise RuntimeError(f"OhemCELoss.forward_upsampled: labels must be int64 (torch.long), got {labels.dtype}")
        if tuple(labels.shape) != (logits_low.shape[0],) + size or labels.device != logits_low.device:
            raise RuntimeError(f"OhemCELoss.forward_upsampled: labels {tuple(labels.shape)} on {labels.device} do not "
                               f"match logits batch {logits_low.shape[0]} x size {size} on {logits_low.device}")
        low = _f32c(logits_low)
        lab = labels.contiguous()
        loss_px, stats = ohem_up_fwd_hip(low.detach(), lab, size, self.thresh, self.ignore_lb)
        return (logits_low, labels, size, low, lab, (loss_px, stats))

    def _fused_finish(self, prep, host_stats):
        logits_low, labels, size, low, lab, fwd = prep
        if fwd is not None:
            from .functional import _OhemUpSelected

            loss_px, stats = fwd
            n_valid, n_above, _ = host_stats if host_stats is not None else stats.tolist()  # host sync
            n_valid, n_above = int(n_valid), int(n_above)
            if n_valid < 0:  # the forward kernel poisons the count when it meets a label outside [0, C) that is not ignore_lb
                raise RuntimeError(f"OhemCELoss.forward_upsampled: label out of range [0, {logits_low.shape[1]}) "
                                   f"(and != ignore_lb {self.ignore_lb}); F.cross_entropy asserts on the same input")
            if n_valid == 0:
                return torch.zeros((), device=logits_low.device, requires_grad=True)
            if n_above >= min(self.n_min, n_valid):
                return _OhemUpSelected.apply(low, lab, loss_px, stats[2], stats[1], size, self.thresh, self.ignore_lb)
        up = F.interpolate(logits_low, size=size, mode="bilinear", align_corners=False)
        return self.forward(up, labels)

    def extra_repr(self):
        return f"thresh={self.thresh}, n_min={self.n_min}, ignore_lb={self.ignore_lb}"


class SoftmaxFocalLoss(nn.Module):
    """Focal loss on softmax probabilities with optional per-class weights -- mirror of reference
    ``src/utils/loss.py:86-128`` (API parity of ``src.utils.loss``; not on the timed path: train.py builds OhemCELoss)."""

    def __init__(self, gamma, weight=None, ignore_lb=255):
        super().__init__()
        self.gamma = gamma
        self.ignore_lb = ignore_lb
        if weight is not None and not isinstance(weight, torch.Tensor):
            weight = torch.tensor(weight, dtype=torch.float32)
        self.register_buffer("weight", weight)

    def forward(self, logits, labels):
        log_prob = F.log_softmax(logits, dim=1)
        focal = (1 - log_prob.exp()) ** self.gamma * log_prob
        w = self.weight if isinstance(self.weight, torch.Tensor) else None
        return F.nll_loss(focal, labels, weight=w, ignore_index=self.ignore_lb)


class _PairPrep:
    """Forward state of the two loss heads between their launch and the host's branch decision."""

    def __init__(self, crits, lows, labels, size, heads=None, pair=None):
        self.crits, self.lows, self.labels, self.size = crits, lows, labels, size
        self.heads = heads    # per-head preps of OhemCELoss._fused_launch (separate launches), or
        self.pair = pair      # (low_a, low_b, labels, loss_px (2,B,H,W), stats (2,3)) of the one-launch form
        # (2,3) device tensor [n_valid, n_above, sum_above] per head, or None when a head is not fused
        if pair is not None:
            self.stats = pair[4]
        elif all(h[5] is not None for h in heads):
            self.stats = torch.stack([h[5][1] for h in heads])
        else:
            self.stats = None


def fused_pair_launch(crit_a, low_a, crit_b, low_b, labels, size):
    """Forward kernels of both loss heads (reference train.py:435), ONE launch when the heads agree in shape, threshold and
    ignore label (they do in the reference's step: two OhemCELoss(0.7, n_min, 255) on two (B,ncls,H/8,W/8) outputs), else one
    launch per head.  No host synchronisation: ``prep.stats`` is a device tensor."""
    size = tuple(size) if size is not None else tuple(labels.shape[-2:])
    same = (low_a.is_cuda and low_b.is_cuda and low_a.shape == low_b.shape and low_a.shape[1] <= 32
            and not isinstance(crit_a.weight, torch.Tensor) and not isinstance(crit_b.weight, torch.Tensor)
            and (crit_a.thresh, crit_a.ignore_lb) == (crit_b.thresh, crit_b.ignore_lb))
    if not same:
        return _PairPrep((crit_a, crit_b), (low_a, low_b), labels, size,
                         heads=(crit_a._fused_launch(low_a, labels, size), crit_b._fused_launch(low_b, labels, size)))
    from .functional import _f32c, ohem_up_pair_fwd_hip

    if labels.dtype != torch.int64:
        raise RuntimeError(f"OhemCELoss.forward_upsampled: labels must be int64 (torch.long), got {labels.dtype}")
    if tuple(labels.shape) != (low_a.shape[0],) + size or labels.device != low_a.device:
        raise RuntimeError(f"OhemCELoss.forward_upsampled: labels {tuple(labels.shape)} on {labels.device} do not "
                           f"match logits batch {low_a.shape[0]} x size {size} on {low_a.device}")
    la, lb_, lab = _f32c(low_a), _f32c(low_b), labels.contiguous()
    loss_px, stats = ohem_up_pair_fwd_hip(la.detach(), lb_.detach(), lab, size, crit_a.thresh, crit_a.ignore_lb)
    return _PairPrep((crit_a, crit_b), (low_a, low_b), labels, size, pair=(la, lb_, lab, loss_px, stats))


def fused_pair_finish(prep, host_stats=None):
    """Loss of both heads from a launched pair; ``host_stats`` = ``prep.stats.tolist()`` if the caller already read it (the
    step's one host sync), else it is read here."""
    crit_a, crit_b = prep.crits
    if prep.pair is None:
        host = host_stats if host_stats is not None else [None, None]
        return crit_a._fused_finish(prep.heads[0], host[0]) + crit_b._fused_finish(prep.heads[1], host[1])
    la, lb_, lab, loss_px, stats = prep.pair
    host = host_stats if host_stats is not None else stats.tolist()  # host sync
    selected = []
    for crit, (n_valid, n_above, _) in zip(prep.crits, host):
        n_valid, n_above = int(n_valid), int(n_above)
        if n_valid < 0:  # the forward kernel poisons the count when it meets a label outside [0, C) that is not ignore_lb
            raise RuntimeError(f"OhemCELoss.forward_upsampled: label out of range [0, {la.shape[1]}) "
                               f"(and != ignore_lb {crit.ignore_lb}); F.cross_entropy asserts on the same input")
        selected.append(n_valid > 0 and n_above >= min(crit.n_min, n_valid))
    if all(selected):
        from .functional import _OhemUpSelectedPair

        return _OhemUpSelectedPair.apply(la, lb_, lab, loss_px, stats, prep.size, crit_a.thresh, crit_a.ignore_lb)
    # a head on the rare top-n_min branch (or without a valid pixel): per-head paths (composite where needed)
    from .functional import _OhemUpSelected

    total = None
    for i, (crit, low, ok) in enumerate(zip(prep.crits, prep.lows, selected)):
        n_valid = int(host[i][0])
        if n_valid == 0:
            term = torch.zeros((), device=low.device, requires_grad=True)
        elif ok:
            term = _OhemUpSelected.apply((la, lb_)[i], lab, loss_px[i], stats[i, 2], stats[i, 1], prep.size, crit.thresh,
                                         crit.ignore_lb)
        else:
            term = crit.forward(F.interpolate(low, size=prep.size, mode="bilinear", align_corners=False), prep.labels)
        total = term if total is None else total + term
    return total


def ohem_upsampled_pair(crit_a, low_a, crit_b, low_b, labels, size):
    """``crit_a.forward_upsampled(low_a, ...) + crit_b.forward_upsampled(low_b, ...)`` with both heads' forward kernels in one
    launch, one backward launch pair, and a single host read-back that decides their OHEM branches (the reference's step,
    train.py:429-441, syncs once per head inside the sort-based loss)."""
    prep = fused_pair_launch(crit_a, low_a, crit_b, low_b, labels, size)
    stats = prep.stats
    return fused_pair_finish(prep, stats.tolist() if stats is not None else None)
