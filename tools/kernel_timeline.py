"""Print the kernel timeline (start offset, duration, stream/queue) of the LAST iteration in a rocprofv3 --kernel-trace
--output-format csv run directory: python tools/kernel_timeline.py <dir> <first-kernel-substring>"""
import csv
import glob
import os
import sys

path = glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2]
starts = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
rows = rows[starts[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1e3:9.1f} .. {e / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  q{r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'][:90]}")
