#!/bin/bash
# In-step kernel durations with a switch at its default / off, SAME box:  bash tools/instep_ab.sh <ENVVAR> <out> [kernel-name regex]
# rocprofv3 --kernel-trace --stats of the eager train steps only (no eval forward, no kernel groups), twice per setting, alternating.
V=${1:-CABINET_WINO_128}; OUT=${2:-$GRAFT_REPO_ROOT/gpurun_out/instep_ab_$V.txt}; RX=${3:-wino}
cd /tmp && export TMPDIR=/tmp
: > $OUT
for val in default 0 default 0; do
  rm -rf /tmp/instep_$val
  if [ $val = default ]; then unset $V; else export $V=$val; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/instep_$val -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-eval-forward --no-graph > /tmp/instep_$val.json 2> /dev/null
  unset $V
  echo "== $V=$val  ($(python3 -c "import json; d=json.loads(open('/tmp/instep_$val.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms/step eager under the profiler')"))" >> $OUT
  python3 - $val "$RX" >> $OUT <<'PY'
import csv, glob, re, sys
f = glob.glob(f"/tmp/instep_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if re.search(sys.argv[2], r["Name"])]
tot = 0.0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    tot += float(r["TotalDurationNs"]) / 1e3 / 15
    print(f'   {r["Name"].split("(")[0][-46:]:46s} calls {r["Calls"]:>4s}  avg {float(r["AverageNs"]) / 1e3:8.1f} us  min {float(r["MinNs"]) / 1e3:8.1f}  max {float(r["MaxNs"]) / 1e3:8.1f}  per step {float(r["TotalDurationNs"]) / 1e3 / 15:8.1f} us')
print(f'   sum of these kernels per step: {tot:8.1f} us')
PY
done
cat $OUT
