#!/usr/bin/env python3
"""Condense the passes of tools/pmc_traffic.sh into profiles-ready JSON: HBM bytes per launch of each kernel group.

For every group and counter two rocprofv3 runs exist (3 and 6 launches); the per-launch value is the difference of the
sums over all `cabinet::` dispatches divided by 3, which removes the operand set-up common to both.  Bytes follow
MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE are in KiB and FETCH_SIZE reports half the bytes of a wide coalesced read
on gfx950, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  The digest of the kernel sources the library was built from
is stored next to the numbers; bench.py refuses the file when the digest differs from the library it runs."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cabinet_amd import build  # noqa: E402


def total(d):
    t, per = 0.0, {}
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "cabinet::" in r["Kernel_Name"]:
                v = float(r["Counter_Value"])
                t += v
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                per[k] = per.get(k, 0.0) + v
    return t, per


def main(root, out_path):
    traffic, detail = {}, {}
    for g in sorted(os.listdir(root)):
        gd = os.path.join(root, g)
        if not os.path.isdir(gd):
            continue
        vals, ok = {}, True
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            t3, p3 = total(os.path.join(gd, c, "3"))
            t6, p6 = total(os.path.join(gd, c, "6"))
            if t6 <= 0 or t3 <= 0:
                ok = False
                break
            vals[c] = (t6 - t3) / 3.0
            detail.setdefault(g, {})[c] = {k: round((p6.get(k, 0.0) - p3.get(k, 0.0)) / 3.0, 1) for k in p6}
        if ok:
            traffic[g] = round((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0)
            detail[g]["FETCH_SIZE_KiB_per_launch"] = round(vals["FETCH_SIZE"], 1)
            detail[g]["WRITE_SIZE_KiB_per_launch"] = round(vals["WRITE_SIZE"], 1)
    out = {"source_digest": build.source_digest(), "batch": int(os.environ.get("CAB_B", "8")),
           "height": int(os.environ.get("CAB_H", os.environ.get("CAB_SIZE", "1024"))),
           "width": int(os.environ.get("CAB_W", os.environ.get("CAB_SIZE", "1024"))),
           "classes": int(os.environ.get("CAB_CLASSES", "8")),
           "method": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes, counters only), 6-launch minus 3-launch sums "
                     "over cabinet:: dispatches / 3; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH calibration)",
           "traffic": traffic, "per_kernel_KiB": detail}
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
    for g, b in traffic.items():
        print(f"{g:16s} {b / 1e6:10.1f} MB per launch")
    print("wrote", out_path)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
