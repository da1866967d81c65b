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
