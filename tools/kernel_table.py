"""Print the per-kernel table of a rocprofv3 --kernel-trace --stats --output-format csv run directory."""
import csv
import glob
import os
import sys

stats = glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1
print(f"{tot / 1e3 / iters:.1f} us of kernel time per iteration ({iters} iterations)")
for r in rows:
    print(f"{float(r['TotalDurationNs']) / 1e3 / iters:8.1f} us/iter  {int(r['Calls']) / iters:5.1f} calls/iter  avg {float(r['AverageNs']) / 1e3:7.1f}  {r['Name'][:120]}")
