"""Run ONE piece of the hot path repeatedly (for rocprofv3 --kernel-trace --stats): python tools/trace_piece.py <piece> [iters]
pieces: global (K6 + K1/K2 + conv1x1), local (K5), block (whole CAB), ffm_up, attn."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cabinet_amd.functional import cab_attention
from cabinet_amd.models.cab import ContextAggregationBlock
from cabinet_amd.models.cabinet import FeatureFusionModule

piece = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B, C, H, W = 8, 256, 32, 32
torch.manual_seed(0)
m = ContextAggregationBlock(C, C // 2).cuda().train()
torch.nn.init.kaiming_normal_(m.global_attn.project_out.weight)
with torch.no_grad():
    m.gamma.fill_(0.5)
x = torch.randn(B, C, H, W, device="cuda", requires_grad=True)
g = torch.randn_like(x)
n = H * W
q, k, v = (torch.randn(B, C // 2, n, device="cuda", requires_grad=True) for _ in range(3))
gc = torch.randn(B, C // 2, n, device="cuda")
ffm = FeatureFusionModule(384, 256).cuda().train()
fsp = torch.randn(B, 128, 4 * H, 4 * W, device="cuda", requires_grad=True)
low = torch.randn(B, 256, H, W, device="cuda", requires_grad=True)
go = torch.randn(B, 256, 4 * H, 4 * W, device="cuda")
fns = {
    "global": lambda: m.global_attn(x).backward(g),
    "local": lambda: m.local_attn(x).backward(g),
    "block": lambda: m(x).backward(g),
    "attn": lambda: cab_attention(q, k, v, (C // 2) ** -0.5).backward(gc),
    "ffm_up": lambda: ffm.forward_upsampled(fsp, low).backward(go),
}
for _ in range(iters):
    x.grad = None
    fns[piece]()
torch.cuda.synchronize()
