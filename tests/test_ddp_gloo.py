"""world_size-2 gloo tests of the data-parallel path (CPU): bucketed all-reduce yields the
rank-average gradient, identical on every rank, equal to a single-process reference."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(4)
    from cabinet_amd.ddp import BucketedGradReducer, init_distributed
    from cabinet_amd.train import TrainStep, build_model, make_criteria, synthetic_batch

    init_distributed("gloo")
    # different seeds per rank: the reducer must broadcast rank 0's weights
    net = build_model("small", n_classes=8, seed=rank, gamma=0.5).train()
    reducer = BucketedGradReducer(net, first_bucket_mb=0.5, bucket_mb=4.0)
    step = TrainStep(net, make_criteria(2, 96, 96, "cpu"), reducer=reducer)
    im, lb = synthetic_batch(2, 96, 96, 8, "cpu", seed=100 + rank)
    losses = [float(step(im, lb)) for _ in range(2)]  # two steps: re-arming of the buckets
    grads = {k: p.grad.clone() for k, p in net.named_parameters() if p.requires_grad}
    torch.save({"grads": grads, "losses": losses, "buckets": reducer.bucket_megabytes,
                "w0": net.sb.conv1.conv.weight.detach().clone()}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_bucketed_allreduce(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"r{i}.pt", weights_only=False) for i in range(world)]
    assert torch.equal(r[0]["w0"], r[1]["w0"])                      # weights were broadcast from rank 0
    assert len(r[0]["buckets"]) >= 3 and abs(sum(r[0]["buckets"]) - 5.36e6 * 4 / 2 ** 20) < 6
    for k in r[0]["grads"]:
        assert torch.equal(r[0]["grads"][k], r[1]["grads"][k]), k   # identical on both ranks
    # single-process reference: mean of the two per-rank gradients (per-rank BN / OHEM, as under DDP)
    from cabinet_amd.train import TrainStep, build_model, make_criteria, synthetic_batch

    ref = None
    for rank in range(world):
        net = build_model("small", n_classes=8, seed=0, gamma=0.5).train()
        step = TrainStep(net, make_criteria(2, 96, 96, "cpu"))
        im, lb = synthetic_batch(2, 96, 96, 8, "cpu", seed=100 + rank)
        step(im, lb)
        step(im, lb)
        g = {k: p.grad for k, p in net.named_parameters() if p.requires_grad}
        ref = g if ref is None else {k: (ref[k] + g[k]) / 2 for k in g}
    for k, want in ref.items():
        got = r[0]["grads"][k]
        err, den = float((got - want).norm()), float(want.norm())
        assert err <= 2e-3 * den + 1e-7, (k, err, den)


def test_reducer_requires_process_group():
    from cabinet_amd.ddp import BucketedGradReducer

    if dist.is_initialized():
        pytest.skip("process group active")
    with pytest.raises(RuntimeError, match="process group"):
        BucketedGradReducer(torch.nn.Linear(2, 2))
