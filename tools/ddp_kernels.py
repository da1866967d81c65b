"""Kernels that only the data-parallel step runs (RCCL, gradient packing): python tools/ddp_kernels.py <rocprof dir> <steps>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True)[0]
steps = int(sys.argv[2])
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6 / steps:.3f} ms per step over {steps} steps")
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("nccl", "Nccl", "rccl", "multi_tensor", "foreach", "copyBuffer", "fill", "Fill", "memset", "direct_copy")):
        print(f"{float(r['TotalDurationNs']) / 1e3 / steps:9.1f} us/step {int(r['Calls']) / steps:7.1f} calls/step avg "
              f"{float(r['AverageNs']) / 1e3:8.1f}  {n[:110]}")
