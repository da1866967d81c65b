#!/bin/bash
# Do the K11 kernels take more CYCLES inside the step than in the replayed loop, or only more time per cycle?
# GRBM_GUI_ACTIVE per launch (counters only, one pass each): (A) the eager train step, (B) tools/run_kernels.py loops of the groups.
#   bash tools/instep_cycles.sh <out>
OUT=${1:-$GRAFT_REPO_ROOT/gpurun_out/instep_cycles.txt}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cyc_step /tmp/cyc_loop_f /tmp/cyc_loop_b
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/cyc_step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-eval-forward --no-graph > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/cyc_loop_f -- python3 $GRAFT_REPO_ROOT/tools/run_kernels.py 8 conv3x3_out_fwd > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/cyc_loop_b -- python3 $GRAFT_REPO_ROOT/tools/run_kernels.py 8 conv3x3_out_bwd > /dev/null 2>&1
python3 - > $OUT <<'PY'
import collections, csv, glob, statistics
def per_kernel(root):
    acc = collections.defaultdict(list)
    for f in glob.glob(root + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "wino" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]) / 8.0)   # summed over 8 XCDs
    return acc
print("GRBM_GUI_ACTIVE / 8 = GPU cycles per launch (counters only; config 3)")
for name, root in (("inside the eager train step", "/tmp/cyc_step"), ("replayed loop, conv3x3_out_fwd", "/tmp/cyc_loop_f"), ("replayed loop, conv3x3_out_bwd", "/tmp/cyc_loop_b")):
    print("==", name)
    for k, v in sorted(per_kernel(root).items()):
        big = [x for x in v if x > 0.5 * max(v)]   # conv_out's launches (the attention branch's are 3-4x shorter)
        print(f"   {k[-44:]:44s} launches {len(v):4d}   median of the long ones ({len(big)}): {statistics.median(big):12.0f} cycles   min {min(big):12.0f}  max {max(big):12.0f}")
PY
cat $OUT
