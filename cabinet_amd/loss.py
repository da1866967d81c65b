"""OHEM cross-entropy -- mirror of reference ``src/utils/loss.py:11-83``.

Not a hand-written kernel (SURVEY.md section 8(f) row f3 lists it as "next"); it is part of
the timed step (BASELINE config 3), so the semantics are restated exactly: per-pixel CE,
drop ignored pixels, full descending sort, keep everything above ``thresh`` if the
``n_min``-th hardest is above it, else the ``n_min`` hardest; mean.
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
        kept = loss[labels != self.ignore_lb]
        if kept.numel() == 0:
            return torch.zeros((), device=logits.device, requires_grad=True)
        ranked, _ = torch.sort(kept, descending=True)
        n_min = min(self.n_min, ranked.numel())
        if ranked[n_min - 1] > self.thresh:
            hard = ranked[ranked > self.thresh]
        else:
            hard = ranked[:n_min]
        return hard.mean()

    def extra_repr(self):
        return f"thresh={self.thresh}, n_min={self.n_min}, ignore_lb={self.ignore_lb}"
