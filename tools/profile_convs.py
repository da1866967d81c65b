"""Which stock convolutions cost what: torch.profiler over two train steps (config 3), device time grouped by op and
input shapes, with FLOPs for a sense of efficiency."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

from cabinet_amd.train import TrainStep, build_model, make_criteria, synthetic_batch

dev = "cuda"
net = build_model("large", n_classes=8, device=dev, seed=0, gamma=0.5).train()
opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=1e-4, momentum=0.9)
step = TrainStep(net, make_criteria(8, 1024, 1024, dev), optimizer=opt)
im, lb = synthetic_batch(8, 1024, 1024, 8, dev)
for _ in range(3):
    step(im, lb)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(2):
        step(im, lb)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if "conv" in e.key.lower() and e.device_time_total > 0:
        rows.append((e.device_time_total / 2, e.key, e.count // 2, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
for t, k, c, sh in rows[:45]:
    print(f"{t:9.1f} us/step  x{c}  {k[:40]:40s} {sh}")
