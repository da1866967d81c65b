"""Stress: hipGraph captures while RCCL's watchdog thread is polling finished collectives (world size 1).
python tools/stress_capture_with_pg.py [global|thread_local] [captures]"""
import os
import socket
import sys

import torch
import torch.distributed as dist

mode = sys.argv[1] if len(sys.argv) > 1 else "thread_local"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0, device_id=torch.device("cuda", 0))
x = torch.randn(1 << 20, device="cuda")
y = torch.zeros_like(x)
for i in range(n):
    for _ in range(8):
        dist.all_reduce(x, async_op=True)  # works for the watchdog to poll
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode=mode):
        for _ in range(50):
            y.add_(1.0)
    g.replay()
torch.cuda.synchronize()
print(f"{mode}: {n} captures with a live process group: ok, y[0] = {float(y[0])}")
dist.destroy_process_group()
