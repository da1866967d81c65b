"""BASELINE config 5: Large, 2x3x2048x1024 (H' x W' = 64 x 32, n = 2048), 19 classes -- one timed train step loop."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cabinet_amd.train import TrainStep, build_model, make_criteria

dev = "cuda"
net = build_model("large", n_classes=19, device=dev, seed=0, gamma=0.5).train()
opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=1e-4, momentum=0.9, weight_decay=5e-4)
H, W, B = 2048, 1024, 2
step = TrainStep(net, make_criteria(B, H, W, dev), optimizer=opt)
g = torch.Generator().manual_seed(1)
im = torch.randn(B, 3, H, W, generator=g).to(dev)
lb = torch.randint(0, 19, (B, H, W), generator=g).to(dev)
for _ in range(3):
    loss = step(im, lb)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    loss = step(im, lb)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"config 5: {B}x3x{H}x{W}, 19 classes: {1e3 * dt / n:.1f} ms/step, {B * n / dt:.1f} images/s, loss {float(loss):.4f}")
