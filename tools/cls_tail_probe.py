import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.nn as nn, torch.nn.functional as F
from torch.profiler import ProfilerActivity, profile
import cabinet_amd.functional as Fh
dev = "cuda"
for (B, C, H, W, K, tag) in ((8, 256, 128, 128, 8, "conv_out tail"), (8, 256, 32, 32, 8, "ab.b2-b4")):
    z = torch.randn(B, C, H, W, device=dev, requires_grad=True)
    bn = nn.BatchNorm2d(C).to(dev).train()
    cls = nn.Conv2d(C, K, 1, bias=(tag != "conv_out tail")).to(dev)
    g = torch.randn(B, K, H, W, device=dev)
    def step():
        z.grad = None
        y = Fh.bn_relu_cls(z, bn, cls)   # CABINET_BN_CLS=0: K7 + the stock 1x1 convolution
        y.backward(g)
    for _ in range(3): step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(5): step()
        torch.cuda.synchronize()
    tot = 0
    print("==", tag)
    for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total):
        if e.device_type.name != "CUDA" and e.self_device_time_total <= 0: continue
        if e.self_device_time_total > 0 and not e.key.startswith(("aten::", "autograd", "_BnAct", "_BnCls", "torch")):
            print(f"{e.self_device_time_total / 5:9.1f} us/step  x{e.count // 5}  {e.key[:100]}")
            tot += e.self_device_time_total / 5
    print(f"   total {tot:.1f} us/step")
