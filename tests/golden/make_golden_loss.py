#!/usr/bin/env python3
"""Golden vectors for ``SoftmaxFocalLoss`` (reference src/utils/loss.py:86-128), produced by running the REFERENCE.

Build container only (needs /root/reference):  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_loss.py
Writes g4_focal.npz: logits (2,19,16,16), labels with ignored pixels, class weights, and for gamma in (0, 1, 2, 5)
with and without weights the loss and dlogits.  Data only.
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
from src.utils.loss import SoftmaxFocalLoss  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
torch.manual_seed(21)
logits = torch.randn(2, 19, 16, 16) * 2
labels = torch.randint(0, 19, (2, 16, 16))
labels[torch.rand(2, 16, 16) < 0.1] = 255
weight = torch.rand(19) + 0.5
out = dict(logits=logits.numpy(), labels=labels.numpy(), weight=weight.numpy(), gammas=np.array([0.0, 1.0, 2.0, 5.0]))
for gi, gamma in enumerate(out["gammas"]):
    for tag, w in (("plain", None), ("weighted", weight)):
        x = logits.clone().requires_grad_(True)
        loss = SoftmaxFocalLoss(float(gamma), weight=w, ignore_lb=255)(x, labels)
        loss.backward()
        out[f"{tag}.{gi}.loss"] = loss.detach().numpy()
        out[f"{tag}.{gi}.dlogits"] = x.grad.numpy()
np.savez_compressed(os.path.join(HERE, "g4_focal.npz"), **out)
print("wrote g4_focal.npz", os.path.getsize(os.path.join(HERE, "g4_focal.npz")))
