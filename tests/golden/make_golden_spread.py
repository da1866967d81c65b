#!/usr/bin/env python3
"""What the REFERENCE itself delivers for full-model gradients: its own run-to-run and precision spread, measured on the imported
reference (not on the oracle restatement) and stored as a fixture, so that the parity rule's yardstick -- "no further from fp64 than
the fp32 reference is" -- is pinned to the real thing ("nothing pins that claim to the real thing").

Build container only (needs /root/reference, which never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_spread.py

CABiNet-MobileNetV3-Large, 2x3x1024x1024, 8 classes, model seed 0, data seed 1, gamma = 0.5, train mode,
fwd + 2x OhemCELoss(0.7, n_min = B*H*W//16) + bwd, three runs of the reference's own modules from the SAME state_dict:
    fp32 with 1 intra-op thread,  fp32 with 8 threads (another summation order inside ATen),  fp64 with 8 threads.
Written to tests/golden/reference_spread_large_2x1024.json: per gradient tensor ||g64||, the relative distances
fp32(1 thread) - fp32(8 threads), fp32(8 threads) - fp64 and fp32(1 thread) - fp64, a projection of the fp64 gradient on a fixed
pseudo-random direction (pins the oracle restatement's fp64 gradients to the reference's at 1e-9 without storing 73 MB), and
the distribution summary (count of tensors past 1e-3, median, p90, max).  Only numbers are written; no reference source text.
"""
import json
import os
import sys

import torch

REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))

import yaml  # noqa: E402
from src.models.cabinet import CABiNet  # noqa: E402
from src.utils.loss import OhemCELoss  # noqa: E402

sys.path.insert(0, HERE)
from make_golden_spread_common import B, NCLS, S, projection_vector, summary  # noqa: E402


def run(sd, cfgs, dtype, threads):
    torch.set_num_threads(threads)
    net = CABiNet(n_classes=NCLS, cfgs=cfgs, mode="large")
    net.load_state_dict(sd)
    net = net.to(dtype).train()
    torch.manual_seed(1)
    x = torch.randn(B, 3, S, S)
    lb = torch.randint(0, NCLS, (B, S, S))
    n_min = B * S * S // 16
    crit_p, crit_16 = OhemCELoss(0.7, n_min, 255), OhemCELoss(0.7, n_min, 255)
    out, out16 = net(x.to(dtype))
    loss = crit_p(out, lb) + crit_16(out16, lb)
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    return float(loss.detach()), grads, out.detach()


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def main():
    with open(f"{REF}/configs/model/mobilenetv3_large.yaml") as f:
        cfgs = yaml.safe_load(f)["cfgs"]
    torch.manual_seed(0)
    net = CABiNet(n_classes=NCLS, cfgs=cfgs, mode="large")
    with torch.no_grad():
        net.ab.a2block.gamma.fill_(0.5)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    del net
    l1, g1, o1 = run(sd, cfgs, torch.float32, 1)
    l8, g8, o8 = run(sd, cfgs, torch.float32, 8)
    l64, g64, o64 = run(sd, cfgs, torch.float64, 8)
    rows = {}
    for k, g in g64.items():
        floor = 1e-7 * g.numel() ** 0.5
        rows[k] = dict(numel=g.numel(), norm64=float(g.norm()), r1_vs_r8=rel(g1[k], g8[k]), r8_vs_f64=rel(g8[k], g),
                       r1_vs_f64=rel(g1[k], g), analytic_zero=bool(float(g.norm()) <= floor),
                       proj64=float((g.flatten() * projection_vector(k, g.numel())).sum()))
    live = [r for r in rows.values() if not r["analytic_zero"]]
    out = {
        "what": "imported reference (dronefreak/CABiNet src.models.cabinet.CABiNet + src.utils.loss.OhemCELoss), Large, "
                f"{B}x3x{S}x{S}, {NCLS} classes, model seed 0, data seed 1, gamma 0.5, train mode, fwd + 2x OHEM-CE + bwd",
        "torch_version": torch.__version__,
        "loss": {"fp32_1thread": l1, "fp32_8threads": l8, "fp64": l64},
        "logits_rel": {"r1_vs_r8": rel(o1, o8), "r8_vs_f64": rel(o8, o64)},
        "summary": {"r1_vs_r8": summary([r["r1_vs_r8"] for r in live]), "r8_vs_f64": summary([r["r8_vs_f64"] for r in live]),
                    "r1_vs_f64": summary([r["r1_vs_f64"] for r in live])},
        "tensors": rows,
    }
    path = os.path.join(HERE, "reference_spread_large_2x1024.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, json.dumps(out["summary"]), out["loss"])


if __name__ == "__main__":
    main()
