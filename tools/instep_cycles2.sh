#!/bin/bash
# launch-by-launch cycles of wino_conv128_kernel inside the eager train step (forward, data gradient alternate)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cyc_step
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/cyc_step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-eval-forward --no-graph > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/cyc_step/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and ("wino_conv128" in r["Kernel_Name"] or "ffm_gate" in r["Kernel_Name"] or "bn_cls_dx" in r["Kernel_Name"]):
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-24:], float(r["Counter_Value"]) / 8.0))
rows.sort()
print(" ".join(f"{n.split('::')[-1][:10]}:{c/1e3:.0f}k" for _, n, c in rows[-40:]))
PY
