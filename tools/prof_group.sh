#!/bin/bash
# per-kernel average durations of one bench.py kernel group: bash tools/prof_group.sh <group> [launches]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pg; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -- python3 $GRAFT_REPO_ROOT/tools/run_kernels.py ${2:-20} $1 > /tmp/pg.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob
f=glob.glob("/tmp/pg/**/*kernel_stats.csv",recursive=True)
print(open("/tmp/pg.log").read()[-800:] if not f else "")
for r in list(csv.DictReader(open(f[0])))[:14]:
    print(r["Name"][:90], r["Calls"], r["AverageNs"])
PY
