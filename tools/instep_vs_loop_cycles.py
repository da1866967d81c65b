#!/usr/bin/env python3
"""Every hand-written kernel's GPU cycles per launch INSIDE the eager train step against the same kernel in bench.py's replayed loop.

    cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/cyc_all -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 \
        --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-eval-forward --no-graph
    python tools/instep_vs_loop_cycles.py /tmp/cyc_all profiles/r06_pmc_counters.json > gpurun_out/r06_instep_vs_loop_cycles.txt

The loop figures are the GRBM_GUI_ACTIVE averages of tools/pmc_counters.sh (per kernel group).  Cycles do not depend on the clock the
run happened to get: a kernel that needs more cycles in the step than in the loop is doing work (or waiting for bytes) there that the
loop does not show.  A kernel launched at several shapes in the step (sg_gemm, bn_cls, wino_conv_kernel ...) is listed with the
launches whose cycle count lies within 15 % of a loop figure."""
import collections
import csv
import glob
import json
import statistics
import sys

root, counters = sys.argv[1], json.load(open(sys.argv[2]))
step = collections.defaultdict(list)
for f in glob.glob(root + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "cabinet::" in r["Kernel_Name"]:
            step[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]) / 8.0)
print(f"{'kernel':58s} {'group (loop)':20s} {'loop cycles':>12s} {'in-step median':>15s} {'launches':>9s} {'in-step / loop':>15s}")
for grp, kernels in sorted(counters["groups"].items()):
    for k, row in sorted(kernels.items()):
        if k not in step or not row.get("GRBM_GUI_ACTIVE") or row["GRBM_GUI_ACTIVE"] / 8.0 < 8000:
            continue
        loop = row["GRBM_GUI_ACTIVE"] / 8.0
        near = [c for c in step[k] if 0.85 * loop <= c <= 1.35 * loop]
        if len(near) < 3:
            continue
        med = statistics.median(near)
        print(f"{k[-58:]:58s} {grp:20s} {loop:12.0f} {med:15.0f} {len(near):9d} {med / loop:15.3f}")
