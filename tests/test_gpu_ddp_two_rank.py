"""GraphedDDPStep with REAL graphs and two ranks: both processes share GPU 0 and talk over gloo (RCCL refuses two ranks on one
device), so what is exercised is the schedule itself -- hipGraph replays with eager collectives on device tensors between
them, the gradient packing into flat buckets, a rank that leaves the graphs for one step (constant-zero loss) while the
other replays -- not the transport."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    torch.cuda.set_device(0)
    from cabinet_amd.train import GraphedDDPStep, build_model, make_criteria, synthetic_batch

    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    for graphs in (True, False):
        net = build_model("small", n_classes=8, seed=rank, gamma=0.5, device="cuda").train()  # rank 0's weights win
        opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=0.01, momentum=0.9)
        step = GraphedDDPStep(net, make_criteria(2, 128, 128, "cuda"), optimizer=opt, bucket_mb=4.0, warmup=1,
                              use_graphs=graphs)
        losses = []
        for i in range(4):  # eager + capture, then three replays
            im, lb = synthetic_batch(2, 128, 128, 8, "cuda", seed=100 + rank + 10 * i)
            losses.append(float(step(im, lb)))
            if i == 1:  # after the first REPLAYED step: two steps from identical weights
                torch.cuda.synchronize()
                w_first_replay = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()
                                  if v.is_floating_point() and "running" not in k}
        im, lb = synthetic_batch(2, 128, 128, 8, "cuda", seed=300 + rank)
        if rank == 1:
            lb = torch.full_like(lb, 255)  # this rank's loss is the constant zero: it leaves the graphs for one step
        losses.append(float(step(im, lb)))
        im, lb = synthetic_batch(2, 128, 128, 8, "cuda", seed=400 + rank)
        losses.append(float(step(im, lb)))  # and both are back on the graphs
        torch.cuda.synchronize()
        res[graphs] = {"losses": losses, "fallbacks": step.fallbacks, "captured": step.graphs is not None,
                       "w1": w_first_replay,
                       "grads": {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters() if p.requires_grad},
                       "w": {k: v.detach().cpu().clone() for k, v in net.state_dict().items()
                             if v.is_floating_point() and "running" not in k}}
        del step, opt, net
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_graphs_and_collectives(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"r{i}.pt", weights_only=False) for i in range(world)]
    for rank in range(world):
        assert r[rank][True]["captured"] and not r[rank][False]["captured"]
    assert r[0][True]["fallbacks"] == 0 and r[1][True]["fallbacks"] == 1
    assert r[1][True]["losses"][4] == 0.0 and r[0][True]["losses"][4] > 0
    for graphs in (True, False):  # replicas stay identical: same averaged gradients, same weights, bit for bit
        for key in ("grads", "w"):
            for k in r[0][graphs][key]:
                assert torch.equal(r[0][graphs][key][k], r[1][graphs][key][k]), (graphs, key, k)
    # the graphed schedule follows the eager one.  Two GPU runs of the stock backward kernels (atomics) differ at the 1e-4
    # level per step in the ill-conditioned first layers and six SGD steps carry that along (seen: 2 % on a 16-element
    # BatchNorm bias of norm 6e-3, 0.6 % on the first convolution at lr = 0.05); a schedule bug -- a bucket not averaged, a
    # stale gradient, a lost segment -- is an O(1) error on every tensor
    for rank in range(world):
        for a, b in zip(r[rank][True]["losses"], r[rank][False]["losses"]):
            assert abs(a - b) <= 2e-3 * max(1.0, abs(b)), (r[rank][True]["losses"], r[rank][False]["losses"])
    for k, w in r[0][False]["w"].items():
        err, den = float((r[0][True]["w"][k].double() - w.double()).norm()), float(w.double().norm())
        assert err <= 3e-2 * den + 1e-4 * w.numel() ** 0.5, (k, err, den)
    # ... and tightly where run-to-run noise has not compounded yet: after the first replayed step (two SGD steps from
    # identical weights) graphed and eager weights agree to 2e-3 per tensor
    for k, w in r[0][False]["w1"].items():
        err, den = float((r[0][True]["w1"][k].double() - w.double()).norm()), float(w.double().norm())
        assert err <= 2e-3 * den + 1e-5 * w.numel() ** 0.5, (k, err, den)
