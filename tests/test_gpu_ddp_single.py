"""RCCL path of the gradient reducer on one GPU (world_size 1): async all_reduce(AVG) launched from the
post-accumulate hooks on bucket views, joined by finish(); gradients must equal a plain backward."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rccl_bucketed_reducer_world1():
    from cabinet_amd.ddp import BucketedGradReducer
    from cabinet_amd.train import TrainStep, build_model, make_criteria, synthetic_batch

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if dist.is_initialized():
        pytest.skip("process group already active")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", world_size=1, rank=0,
                            device_id=torch.device("cuda", 0))
    try:
        im, lb = synthetic_batch(2, 128, 128, 8, "cuda", seed=3)
        ref = build_model("small", n_classes=8, seed=0, gamma=0.5, device="cuda").train()
        TrainStep(ref, make_criteria(2, 128, 128, "cuda"))(im, lb)
        net = build_model("small", n_classes=8, seed=0, gamma=0.5, device="cuda").train()
        reducer = BucketedGradReducer(net, first_bucket_mb=0.5, bucket_mb=4.0, always_reduce=True)
        assert reducer.backend == "nccl" and len(reducer.buckets) >= 3
        step = TrainStep(net, make_criteria(2, 128, 128, "cuda"), reducer=reducer)
        step(im, lb)
        step(im, lb)  # buckets re-arm and re-zero
        torch.cuda.synchronize()
        for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
            if p.requires_grad:
                assert p.grad.data_ptr() >= reducer._bucket_of[p].flat.data_ptr()
                # two separate GPU runs: stock PyTorch-ROCm backward kernels use atomics (upsample, indexing),
                # so ill-conditioned backbone tensors differ run to run at the 1e-4..1e-3 level; a reducer bug
                # (missing average, detached view, stale bucket) would be an O(1) error
                err, den = float((p.grad - q.grad).norm()), float(q.grad.norm())
                assert err <= 2e-2 * den + 1e-6 * p.numel() ** 0.5, (k, err, den)
    finally:
        dist.destroy_process_group()


def test_graphed_ddp_step_world1_matches_graphed_single():
    """GraphedDDPStep on one GPU with RCCL initialised and the collectives forced (world 1): four hipGraph segments with
    eager all_reduce(AVG) calls between them, captured while the process group (and its watchdog thread) is alive.
    Losses and weights must follow GraphedTrainStep step for step; a batch with every label ignored takes the eager
    fallback (same collectives) without disturbing BatchNorm buffers."""
    from cabinet_amd.train import GraphedDDPStep, GraphedTrainStep, build_model, make_criteria, synthetic_batch

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if dist.is_initialized():
        pytest.skip("process group already active")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", world_size=1, rank=0,
                            device_id=torch.device("cuda", 0))
    try:
        batches = [synthetic_batch(2, 256, 256, 8, "cuda", seed=30 + i) for i in range(4)]
        ign = (batches[0][0], torch.full_like(batches[0][1], 255))
        res = []
        for ddp in (False, True):
            net = build_model("small", n_classes=8, seed=0, gamma=0.5, device="cuda").train()
            # plain SGD: on the all-ignored batch the data-parallel step applies an (all-reduced) ZERO gradient where the
            # single-process step has no gradient at all -- identical without momentum, which is all this test is about
            opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=1e-2)
            crit = make_criteria(2, 256, 256, "cuda")
            step = (GraphedDDPStep(net, crit, optimizer=opt, warmup=1, always_reduce=True, bucket_mb=4.0) if ddp
                    else GraphedTrainStep(net, crit, optimizer=opt, warmup=1))
            losses = [float(step(*b)) for b in batches] + [float(step(*ign))] + [float(step(*batches[1]))]
            if ddp:
                assert step.graphs is not None and step.fallbacks == 1 and len(step.bucket_megabytes) >= 2
            res.append((losses, {k: v.clone() for k, v in net.state_dict().items()}))
        (la, sa), (lb_, sb) = res
        for x, y in zip(la, lb_):
            assert abs(x - y) <= 1e-5 * max(1.0, abs(x)), (la, lb_)
        for k in sa:
            # two GPU runs of stock backward kernels with atomics differ at the 1e-4 level per step and SGD carries it
            # along; a schedule bug (missing average, stale bucket, lost segment) is an O(1) error
            err, den = float((sb[k].double() - sa[k].double()).norm()), float(sa[k].double().norm())
            assert err <= 2e-3 * den + 1e-5 * sa[k].numel() ** 0.5, (k, err, den)
    finally:
        dist.destroy_process_group()


def test_graph_capture_with_live_rccl_watchdog():
    """RCCL's watchdog thread polls the events of finished collectives; under torch's default (global) capture error mode
    such a hipEventQuery while ANY capture is open terminates the process ("operation not permitted when stream is
    capturing" -- tools/stress_capture_with_pg.py global reproduces it within a few captures).  Every capture of this
    package uses cabinet_amd.train._CAPTURE_MODE; 100 captures with collectives in flight before each must survive."""
    from cabinet_amd.train import _CAPTURE_MODE

    assert _CAPTURE_MODE == "thread_local"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if dist.is_initialized():
        pytest.skip("process group already active")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", world_size=1, rank=0,
                            device_id=torch.device("cuda", 0))
    try:
        x = torch.randn(1 << 18, device="cuda")
        y = torch.zeros(1 << 10, device="cuda")
        for _ in range(100):
            for _ in range(8):
                dist.all_reduce(x, async_op=True)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=_CAPTURE_MODE):
                for _ in range(20):
                    y.add_(1.0)
            g.replay()
        torch.cuda.synchronize()
        assert float(y[0]) == 100 * 20  # a capture records, only the replay executes
    finally:
        dist.destroy_process_group()
