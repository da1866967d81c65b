"""Per-step launch counts of the small (< 8 us average) kernels of a rocprofv3 --kernel-trace --stats run:
python tools/small_kernels.py <dir> <steps>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True)[0]
steps = int(sys.argv[2])
rows = [r for r in csv.DictReader(open(f)) if float(r["AverageNs"]) < 8000]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"small kernels: {tot / 1e3 / steps:.0f} us per step, {sum(int(r['Calls']) for r in rows) / steps:.0f} launches per step")
for r in rows[:25]:
    print(f"{float(r['TotalDurationNs']) / 1e3 / steps:8.1f} us/step {int(r['Calls']) / steps:7.1f} calls/step avg "
          f"{float(r['AverageNs']) / 1e3:6.1f}  {r['Name'][:120]}")
