#!/usr/bin/env python3
"""Launch ONE hand-written kernel group of bench.py `iters` times (for rocprofv3 passes) at the workload shape the environment
names: CAB_B x 3 x CAB_H x CAB_W, CAB_CLASSES classes (defaults = BASELINE config 3: 8 x 3 x 1024 x 1024, 8 classes;
config 5: CAB_B=2 CAB_H=2048 CAB_W=1024 CAB_CLASSES=19).

    python tools/run_kernels.py <iters> <group>      group = first word of a bench.py kernel name, e.g. cab_attn_fwd

The groups, their operands and launch closures are bench.kernel_cases(): what is profiled is exactly what bench.py times.
Operand setup of the groups in front of the requested one also runs (once); tools/summarize_traffic.py removes every
such constant by differencing two runs with different `iters`."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
which = sys.argv[2] if len(sys.argv) > 2 else "all"
B = int(os.environ.get("CAB_B", "8"))
H = int(os.environ.get("CAB_H", os.environ.get("CAB_SIZE", "1024")))
W = int(os.environ.get("CAB_W", os.environ.get("CAB_SIZE", "1024")))
ncls = int(os.environ.get("CAB_CLASSES", "8"))
found = False
for name, fn, flops, nbytes, bound in bench.kernel_cases(B, H, W, ncls, extra=True):
    key = name.split(" ")[0]
    if which in ("all", key):
        found = True
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        if which != "all":
            break
if not found:
    raise SystemExit(f"unknown kernel group {which!r}")
