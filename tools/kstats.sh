#!/bin/bash
# per-kernel average durations of one bench.py kernel group under rocprofv3 (on the GPU box):  bash tools/kstats.sh <group> [iters]
G=${1:-ffm_up_bwd}; N=${2:-20}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kstats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -- python3 $GRAFT_REPO_ROOT/tools/run_kernels.py $N $G > /tmp/kstats.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/kstats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if int(r["Calls"]) >= 10:
        print(f'{r["Name"].split("(")[0][-48:]:48s} calls {r["Calls"]:>4s}  avg {float(r["AverageNs"]) / 1e3:8.1f} us  min {float(r["MinNs"]) / 1e3:8.1f}')
PY
