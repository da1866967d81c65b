"""How long the host needs to enqueue one training step vs how long the GPU needs to run it."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cabinet_amd.train import TrainStep, build_model, make_criteria, synthetic_batch

dev = "cuda"
net = build_model("large", n_classes=8, device=dev, seed=0, gamma=0.5).train()
opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=1e-4, momentum=0.9, weight_decay=5e-4)
step = TrainStep(net, make_criteria(8, 1024, 1024, dev), optimizer=opt)
im, lb = synthetic_batch(8, 1024, 1024, 8, dev)
for _ in range(5):
    step(im, lb)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    step(im, lb)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / n:.1f} ms/step; wall {1e3 * (t2 - t0) / n:.1f} ms/step "
      f"(the OHEM branch decision reads two scalars back per head, so the host cannot run far ahead)")

# pure host cost: enqueue without any read-back (dummy loss), GPU queue absorbs the work
def dummy():
    for p in net.parameters():
        p.grad = None
    low, low16 = net.forward_lowres(im)
    (low.sum() + low16.sum()).backward()


for _ in range(3):
    dummy()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    dummy()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"no read-back: host enqueue {1e3 * (t1 - t0) / 5:.1f} ms/step, wall {1e3 * (t2 - t0) / 5:.1f} ms/step")
with torch.no_grad():
    for _ in range(3):
        net.forward_lowres(im)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        net.forward_lowres(im)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"forward only (no_grad): host enqueue {1e3 * (t1 - t0) / 5:.1f} ms, wall {1e3 * (t2 - t0) / 5:.1f} ms")
