"""The CAB's launch chain piece by piece (HIP events over graph replays, as bench.py times its kernel groups):
    python tools/time_cab_chain.py [B H W]        default 8 32 32 (BASELINE config 3's CAB grid); config 5: 2 64 32
rows: K1 alone, the output projection alone (small GEMM), K1 with the projection in its epilogue (where the fused form applies),
the producers (K6) forward / backward, the local branch (K5), and the whole block forward with the fused / two-launch form."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import cabinet_amd.functional as Fh
from bench import time_kernel
from cabinet_amd.models.cab import ContextAggregationBlock


def main():
    B, H, W = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 32, 32)
    C, Kc, n = 256, 128, H * W
    dev = "cuda"
    g = torch.Generator().manual_seed(3)
    q = torch.randn(B, Kc, n, generator=g).relu().to(dev)
    k = torch.randn(B, Kc, n, generator=g).to(dev)
    v = torch.randn(B, Kc, n, generator=g).to(dev)
    w = (torch.randn(C, Kc, generator=g) * Kc ** -0.5).to(dev)
    scale = Kc ** -0.5
    it = 60
    rows = []
    ctx, _ = Fh.attn_fwd_hip(q, k, v, scale)
    ctx4 = ctx.reshape(B, Kc, n, 1)
    rows.append(("K1 alone (cabinet_cab_attn_fwd)", time_kernel(lambda: Fh.attn_fwd_hip(q, k, v, scale), it)))
    with torch.no_grad():
        rows.append(("project_out alone (cabinet_conv1x1_fwd, small GEMM)", time_kernel(lambda: Fh.conv1x1(ctx4, w), it)))
        fused = Fh.cab_attention_proj_supported(q, v, w)
        if fused:
            rows.append(("K1 + project_out in its epilogue, ctx not written (inference)",
                         time_kernel(lambda: Fh.cab_attention_proj(q, k, v, w, scale), it)))
    if fused:
        qg = q.clone().requires_grad_(True)
        rows.append(("K1 + project_out in its epilogue, ctx written (training)",
                     time_kernel(lambda: Fh.cab_attention_proj(qg, k, v, w, scale), it)))
    else:
        print(f"(B={B}, n={n}): the forward splits the keys -- the fused projection does not apply, the block runs K1 + small GEMM")
    with Fh.batched_bn_counters():
        cab = ContextAggregationBlock(C, Kc).to(dev).train()
        torch.nn.init.kaiming_normal_(cab.global_attn.project_out.weight)
        with torch.no_grad():
            cab.gamma.fill_(0.5)
        x = torch.randn(B, C, H, W, generator=g).to(dev).requires_grad_(True)
        q3 = Fh.cab_qkv(x, cab.global_attn)
        gq = [torch.randn_like(t) for t in q3]
        rows.append(("K6 forward (producers)", time_kernel(lambda: Fh.cab_qkv(x.detach(), cab.global_attn), it)))
        rows.append(("K6 backward", time_kernel(lambda: Fh._CabQkv.backward(q3[0].grad_fn, *gq), it)))
        rows.append(("K5 forward (local branch + combine)", time_kernel(lambda: cab.local_attn(x.detach()), it)))
        for flag in (True, False):
            Fh.PROJ_FUSED = flag
            rows.append((f"whole block forward, training graph built, PROJ_FUSED={int(flag)}", time_kernel(lambda: cab(x), it)))
        Fh.PROJ_FUSED = True
    for name, ms in rows:
        print(f"{name:75s} {ms * 1e3:8.1f} us", flush=True)


if __name__ == "__main__":
    main()
