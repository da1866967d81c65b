"""How long the host needs to enqueue one training step vs how long the GPU needs to run it (config 3, one GPU):
eager TrainStep, GraphedTrainStep (two hipGraphs around the one read-back) and, with RCCL forced at world size 1, the
eager hook reducer vs GraphedDDPStep.  Host time = wall time of the calls that enqueue a step, with the step's one
read-back excluded where it can be (the dummy-loss rows) and included where it cannot (the real step: the host must wait
for the OHEM counts).  Written to profiles/ as evidence for DESIGN.md section 6."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cabinet_amd.train import GraphedDDPStep, GraphedTrainStep, TrainStep, build_model, make_criteria, synthetic_batch

dev = "cuda"
im, lb = synthetic_batch(8, 1024, 1024, 8, dev)


def fresh():
    net = build_model("large", n_classes=8, device=dev, seed=0, gamma=0.5).train()
    opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=1e-4, momentum=0.9, weight_decay=5e-4)
    return net, opt, make_criteria(8, 1024, 1024, dev)


def measure(step, label, n=20, warm=5):
    for _ in range(warm):
        step(im, lb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step(im, lb)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{label:58s} host calls return after {1e3 * (t1 - t0) / n:6.2f} ms/step; wall {1e3 * (t2 - t0) / n:6.2f} ms/step", flush=True)


net, opt, crit = fresh()
measure(TrainStep(net, crit, optimizer=opt), "eager TrainStep (incl. waiting for the OHEM read-back)")


def dummy():  # pure enqueue cost of the eager path: no read-back, the GPU queue absorbs the work
    for p in net.parameters():
        p.grad = None
    low, low16 = net.forward_lowres(im)
    (low.sum() + low16.sum()).backward()
    opt.step()


measure(lambda a, b: dummy(), "eager fwd + bwd + SGD, dummy loss (host enqueue only)", n=10, warm=3)
del net, opt
torch.cuda.empty_cache()
net, opt, crit = fresh()
g = GraphedTrainStep(net, crit, optimizer=opt)
measure(g, "GraphedTrainStep (2 graph launches + read-back)")
t0 = time.perf_counter()
for _ in range(20):
    g.s_im.copy_(im, non_blocking=True)
    g.g_fwd.replay()
    g.g_bwd.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"{'GraphedTrainStep without the read-back (host enqueue only)':58s} host calls return after {1e3 * (t1 - t0) / 20:6.2f} ms/step", flush=True)
del net, opt, g
torch.cuda.empty_cache()

if os.environ.get("CABINET_FORCE_DDP") == "1":
    from cabinet_amd.ddp import BucketedGradReducer, init_distributed

    init_distributed()
    net, opt, crit = fresh()
    red = BucketedGradReducer(net, always_reduce=True)
    measure(TrainStep(net, crit, reducer=red, optimizer=opt), "world 1 with RCCL forced: eager TrainStep + hook reducer")
    red.remove()
    del net, opt, red
    torch.cuda.empty_cache()
    net, opt, crit = fresh()
    gd = GraphedDDPStep(net, crit, optimizer=opt, always_reduce=True)
    measure(gd, f"world 1 with RCCL forced: GraphedDDPStep (4 graphs + {len(gd.bucket_megabytes)} all-reduces)")
    torch.distributed.destroy_process_group()
