"""Where does GraphedDDPStep spend what GraphedTrainStep does not, and do its collectives run BESIDE the backward?  (world size 1, RCCL
forced; rocprofv3's kernel trace cannot answer the second question -- under it kernels execute in host submission order across
streams, tools/stream_overlap_probe.py shows they do not without it.)

    CABINET_FORCE_DDP=1 python tools/ddp_segments.py
rows: wall time per step of (a) GraphedTrainStep, (b) GraphedDDPStep with its collectives, (c) the same without them (segmentation and
packing alone), (d) with a 0.5 ms spin kernel issued beside the decoder's collectives on their side stream: if the schedule overlaps,
(d) - (b) is ~0; and HIP-event times of the three backward graphs of (b)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cabinet_amd.ddp import init_distributed
from cabinet_amd.train import GraphedDDPStep, GraphedTrainStep, build_model, make_criteria, synthetic_batch

dev = "cuda"
init_distributed()
im, lb = synthetic_batch(8, 1024, 1024, 8, dev)


def fresh():
    net = build_model("large", n_classes=8, device=dev, seed=0, gamma=0.5).train()
    opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=1e-4, momentum=0.9, weight_decay=5e-4)
    return net, opt, make_criteria(8, 1024, 1024, dev)


def wall(step, n=30, warm=6):
    for _ in range(warm):
        step(im, lb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step(im, lb)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


net, opt, crit = fresh()
g1 = GraphedTrainStep(net, crit, optimizer=opt)
a = wall(g1)
print(f"(a) GraphedTrainStep                                        {a:7.3f} ms/step", flush=True)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
tot = [0.0] * 3
for _ in range(10):
    g1.s_im.copy_(im), g1.s_lb.copy_(lb)
    ev[0].record(); g1.g_fwd.replay(); ev[1].record(); g1.g_bwd.replay(); ev[2].record(); g1.opt_seg.run(); ev[3].record()
    torch.cuda.synchronize()
    for i in range(3):
        tot[i] += ev[i].elapsed_time(ev[i + 1]) / 10
print("    its graphs back to back:  A %.3f  B %.3f  optimizer (eager, enqueued behind B) %.3f ms" % tuple(tot), flush=True)
del net, opt, g1
torch.cuda.empty_cache()

net, opt, crit = fresh()
gd = GraphedDDPStep(net, crit, optimizer=opt, always_reduce=True)
b = wall(gd)
print(f"(b) GraphedDDPStep, {len(gd.bucket_megabytes)} all-reduces (RCCL, world 1)            {b:7.3f} ms/step   (+{b - a:.3f})", flush=True)
gd.always_reduce = False
c = wall(gd)
print(f"(c) the same without the collectives (segments + packing)   {c:7.3f} ms/step   (+{c - a:.3f})", flush=True)
gd.always_reduce = True
# (d) a payload beside the decoder's collectives: ONE kernel that spins ~0.5 ms on a single thread (torch.cuda._sleep: no memory
# traffic, one wave), issued on the stream the collectives are issued on, behind B1's event.  Overlapped: it costs the step nothing;
# serialised behind the backward: its full length.
cycles = int(0.5e-3 * 2.0e9)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
s.record()
torch.cuda._sleep(cycles)
e.record()
torch.cuda.synchronize()
payload_ms = s.elapsed_time(e)
orig_reduce = gd._reduce


def reduce_with_payload(seg):
    works = orig_reduce(seg)
    if seg == 0:
        torch.cuda._sleep(cycles)
    return works


gd._reduce = reduce_with_payload
d = wall(gd)
print(f"(d) (b) + a {payload_ms:.2f} ms spin kernel beside the decoder's collectives   {d:7.3f} ms/step   (+{d - b:.3f} over (b): "
      f"{'OVERLAPPED with the backward' if d - b < 0.35 * payload_ms else 'serialised' if d - b > 0.8 * payload_ms else 'partly overlapped'})", flush=True)
gd._reduce = orig_reduce
# HIP-event times of the backward graphs
gA, gB1, gB2, gB3 = gd.graphs
ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
tot = [0.0] * 4
for _ in range(10):
    gd.s_im.copy_(im), gd.s_lb.copy_(lb)
    gd.snap.save()
    ev[0].record(); gA.replay(); ev[1].record(); gB1.replay(); ev[2].record(); gB2.replay(); ev[3].record(); gB3.replay(); ev[4].record()
    torch.cuda.synchronize()
    gd.snap.restore()
    for i in range(4):
        tot[i] += ev[i].elapsed_time(ev[i + 1]) / 10
print("graphs back to back (no collectives, no optimizer):  A %.3f  B1 %.3f  B2 %.3f  B3 %.3f  sum %.3f ms" % (*tot, sum(tot)), flush=True)
# host timeline of one replayed step (ms after the read-back): is the host ahead of the GPU when it reaches the optimizer?
import cabinet_amd.train as T

stamps = []
orig_join, orig_run = gd._join, gd.opt_seg.run
gd._join = lambda w: (stamps.append(("reduces issued", time.perf_counter())), orig_join(w), stamps.append(("joined", time.perf_counter())))[1]
gd.opt_seg.run = lambda: (orig_run(), stamps.append(("optimizer enqueued", time.perf_counter())))[0]
for _ in range(3):
    stamps.clear()
    t0 = time.perf_counter()
    gd(im, lb)
    t_ret = time.perf_counter()
    torch.cuda.synchronize()
    t_end = time.perf_counter()
print("host timeline of one step (ms from the call): " + ", ".join(f"{n} {1e3 * (t - t0):.2f}" for n, t in stamps)
      + f", call returns {1e3 * (t_ret - t0):.2f}, GPU done {1e3 * (t_end - t0):.2f}", flush=True)
torch.distributed.destroy_process_group()
