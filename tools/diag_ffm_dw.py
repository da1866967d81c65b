"""dW_blk of the fused-upsample FFM backward, per 64-column block, both implementations against the fp64 product."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cabinet_amd import functional as Fn

B, H, W, Hl, Wl = [int(x) for x in sys.argv[1:6]] if len(sys.argv) > 5 else (3, 32, 96, 8, 24)
gen = torch.Generator().manual_seed(1)
Cs, Cc, Co, Cm = 128, 256, 256, 64
fsp = torch.randn(B, Cs, H, W, generator=gen).cuda()
low = torch.randn(B, Cc, Hl, Wl, generator=gen).cuda()
wb = (torch.randn(Co, Cs + Cc, generator=gen) * 0.07).cuda()
w1 = (torch.randn(Cm, Co, generator=gen) * 0.1).cuda()
w2 = (torch.randn(Co, Cm, generator=gen) * 0.1).cuda()
g = torch.randn(B, Co, H, W, generator=gen).cuda()
bw, bb = torch.ones(Co).cuda(), torch.zeros(Co).cuda()
rm, rv = torch.zeros(Co).cuda(), torch.ones(Co).cuda()
out, z, mean, invstd, pooled, gate = Fn.ffm_up_fwd_hip(fsp, low, wb, bw, bb, rm, rv, w1, w2, True, 0.1, 1e-5)
args = (g, fsp, low, wb, bw, bb, w1, w2, z, mean, invstd, pooled, gate, True)
res = {}
for name, env in (("fused", None), ("chain", "1")):
    if env:
        os.environ["CABINET_FFM_BWD_UNFUSED"] = env
    res[name] = [t.double().cpu() for t in Fn.ffm_up_bwd_hip(*args)]
    os.environ.pop("CABINET_FFM_BWD_UNFUSED", None)
torch.cuda.synchronize()
for i, nm in enumerate(("dfsp", "dlow", "dw_blk")):
    a, b = res["fused"][i], res["chain"][i]
    print(nm, "fused vs chain rel", float((a - b).norm() / b.norm()))
a, b = res["fused"][2], res["chain"][2]
for c0 in range(0, Cs + Cc, 64):
    d = (a[:, c0:c0 + 64] - b[:, c0:c0 + 64])
    rows = [float(d[r0:r0 + 64].norm() / b[r0:r0 + 64, c0:c0 + 64].norm()) for r0 in range(0, Co, 64)]
    print("cols", c0, ["%.1e" % r for r in rows])
