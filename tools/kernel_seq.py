"""Print the dispatch sequence (duration, gap to the previous kernel's end) of the LAST `count` kernels of a rocprofv3
--kernel-trace --output-format csv run directory."""
import csv
import glob
import os
import sys

trace = glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-count:]
prev_end = None
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"t={(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  grid {r.get('Grid_Size_X', '?'):>8}  {r['Kernel_Name'][:90]}")
    prev_end = e
print(f"span {(prev_end - t0) / 1e3:.1f} us")
