#!/usr/bin/env python3
"""Condense the passes of tools/pmc_counters.sh: per kernel group -> per kernel -> per-launch counter averages, plus the derived
figures DESIGN.md quotes (MFMA-pipe busy share of the kernel, VALU / SALU / LDS instructions per wave, LDS bank conflicts).

SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs that ran MFMAs; GRBM_GUI_ACTIVE is reported summed over the 8
XCDs (MI355X_MICROARCH.md), so  mfma_busy_share = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8).
The digest of the kernel sources the library was built from is stored next to the numbers."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cabinet_amd import build  # noqa: E402


def main(root, out_path):
    groups = {}
    for g in sorted(os.listdir(root)):
        gd = os.path.join(root, g)
        if not os.path.isdir(gd):
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(gd, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "cabinet::" in r["Kernel_Name"]:
                    acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
        kernels = {}
        for k, cs in acc.items():
            row = {c: round(sum(v) / len(v), 1) for c, v in cs.items()}
            row["launches_seen"] = max(len(v) for v in cs.values())
            if row["launches_seen"] < 4:  # operand set-up of an earlier group (run once or twice), not this group's 4 iterations
                continue
            waves = row.get("SQ_WAVES", 0.0)
            if row.get("GRBM_GUI_ACTIVE"):
                row["mfma_busy_share"] = round(row.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4 * 256 * row["GRBM_GUI_ACTIVE"] / 8.0), 4)
            wc = row.get("SQ_WAVE_CYCLES", 0.0)
            if wc:  # where a wave's lifetime goes (quad-cycle units cancel): issuing / stalled at issue / parked at a waitcnt or barrier
                for c, name in (("SQ_ACTIVE_INST_ANY", "share_issuing"), ("SQ_WAIT_INST_ANY", "share_issue_stall"),
                                ("SQ_WAIT_ANY", "share_waitcnt_or_barrier")):
                    if c in row:
                        row[name] = round(row[c] / wc, 4)
            if waves:
                for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
                    if c in row:
                        row[c + "_per_wave"] = round(row[c] / waves, 1)
            kernels[k] = row
        groups[g] = kernels
    out = {"source_digest": build.source_digest(), "batch": int(os.environ.get("CAB_B", "8")),
           "height": int(os.environ.get("CAB_H", os.environ.get("CAB_SIZE", "1024"))),
           "width": int(os.environ.get("CAB_W", os.environ.get("CAB_SIZE", "1024"))),
           "classes": int(os.environ.get("CAB_CLASSES", "8")),
           "method": "rocprofv3 --pmc, four separate counter-only passes per kernel group over tools/run_kernels.py 4 <group> "
                     "(tools/pmc_counters.sh); per-launch averages per kernel; kernels launched fewer than 4 times in a group's "
                     "run (operand set-up of earlier groups) are dropped",
           "groups": groups}
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
    for g, ks in groups.items():
        for k, row in ks.items():
            print(f"{g:22s} {k[:60]:60s} mfma_busy {row.get('mfma_busy_share', 0):6.3f}  valu/wave {row.get('SQ_INSTS_VALU_per_wave', 0):8.1f}  "
                  f"lds conflicts {row.get('SQ_LDS_BANK_CONFLICT', 0):10.0f}")
    print("wrote", out_path)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
