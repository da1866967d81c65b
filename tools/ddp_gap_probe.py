"""What separates GraphedDDPStep (no collectives) from the sum of its graphs?  (world size 1, RCCL forced)

    CABINET_FORCE_DDP=1 python tools/ddp_gap_probe.py
Wall time per step of the replay path rebuilt by hand from the step's own graphs, adding one ingredient at a time:
  v0  A, B1, B2, B3 replayed back to back, no host read-back, no optimizer                  (the GPU's own time)
  v1  + the host read-back of the OHEM statistics between A and B1                          (the step's one sync)
  v2  + the gradient-view re-binding loop between the read-back and B1
  v3  + event records behind B1 / B2 and the side stream waiting for them
  v4  + the optimizer (eager) behind B3                                                     (= the step without collectives)
and the step itself."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cabinet_amd.ddp import init_distributed
from cabinet_amd.train import GraphedDDPStep, build_model, make_criteria, synthetic_batch

dev = "cuda"
init_distributed()
im, lb = synthetic_batch(8, 1024, 1024, 8, dev)
net = build_model("large", n_classes=8, device=dev, seed=0, gamma=0.5).train()
opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=1e-4, momentum=0.9, weight_decay=5e-4)
gd = GraphedDDPStep(net, make_criteria(8, 1024, 1024, dev), optimizer=opt, always_reduce=False)
for _ in range(6):
    gd(im, lb)
torch.cuda.synchronize()
gA, gB1, gB2, gB3 = gd.graphs


def variant(level):
    def run():
        gd.s_im.copy_(im, non_blocking=True)
        gd.s_lb.copy_(lb, non_blocking=True)
        gd.snap.save()
        gA.replay()
        if level >= 1:
            gd.s_stats.tolist()
        if level >= 2:
            for views in gd.seg_views:
                for p, view in views:
                    p.grad = view
        cur = torch.cuda.current_stream()
        gB1.replay()
        if level >= 3:
            gd._ev[0].record(cur)
        gB2.replay()
        if level >= 3:
            gd._ev[1].record(cur)
        gB3.replay()
        if level >= 3:
            with torch.cuda.stream(gd._side):
                gd._side.wait_event(gd._ev[0])
                gd._side.wait_event(gd._ev[1])
        if level >= 4:
            gd.opt_seg.run()
    return run


def wall(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


names = ["v0 graphs back to back", "v1 + read-back between A and B1", "v2 + gradient views re-bound", "v3 + events / side-stream waits",
         "v4 + optimizer"]
prev = None
for lv, name in enumerate(names):
    t = wall(variant(lv))
    print(f"{name:40s} {t:7.3f} ms/step" + (f"   (+{t - prev:.3f})" if prev is not None else ""), flush=True)
    prev = t
# which part of v3 costs: the records, the waits, one event or two?
def v3_variant(rec0, rec1, wait0, wait1):
    def run():
        gd.s_im.copy_(im, non_blocking=True)
        gd.s_lb.copy_(lb, non_blocking=True)
        gd.snap.save()
        gA.replay()
        gd.s_stats.tolist()
        cur = torch.cuda.current_stream()
        gB1.replay()
        if rec0:
            gd._ev[0].record(cur)
        gB2.replay()
        if rec1:
            gd._ev[1].record(cur)
        gB3.replay()
        if wait0 or wait1:
            with torch.cuda.stream(gd._side):
                if wait0:
                    gd._side.wait_event(gd._ev[0])
                if wait1:
                    gd._side.wait_event(gd._ev[1])
    return run


for name, args in (("records only (2)", (1, 1, 0, 0)), ("record behind B1 only", (1, 0, 0, 0)), ("record behind B2 only", (0, 1, 0, 0)),
                   ("record + wait, B1 only", (1, 0, 1, 0)), ("records + waits (2) = v3", (1, 1, 1, 1))):
    print(f"   {name:36s} {wall(v3_variant(*args)):7.3f} ms/step", flush=True)
side2 = torch.cuda.Stream(device=dev)


def v3_two_streams(only_second):
    def run():
        gd.s_im.copy_(im, non_blocking=True)
        gd.s_lb.copy_(lb, non_blocking=True)
        gd.snap.save()
        gA.replay()
        gd.s_stats.tolist()
        cur = torch.cuda.current_stream()
        gB1.replay()
        gd._ev[0].record(cur)
        gB2.replay()
        gd._ev[1].record(cur)
        gB3.replay()
        if not only_second:
            gd._side.wait_event(gd._ev[0])
        side2.wait_event(gd._ev[1])
    return run


print(f"   {'wait for the B2 event only':36s} {wall(v3_variant(1, 1, 0, 1)):7.3f} ms/step", flush=True)
print(f"   {'the two waits on two streams':36s} {wall(v3_two_streams(False)):7.3f} ms/step", flush=True)
print(f"   {'B2 wait alone on a fresh stream':36s} {wall(v3_two_streams(True)):7.3f} ms/step", flush=True)
t = wall(lambda: gd(im, lb))
print(f"{'the step itself (no collectives)':40s} {t:7.3f} ms/step   (+{t - prev:.3f} over v4)", flush=True)
torch.distributed.destroy_process_group()
