#!/bin/bash
# In-step durations of the K11 kernels with a switch on / off, SAME box:  bash tools/instep_ab.sh <ENVVAR> [out]
# rocprofv3 --kernel-trace --stats of the eager train steps only (no eval forward, no kernel groups), per setting.
V=${1:-CABINET_WINO_128}; OUT=${2:-$GRAFT_REPO_ROOT/gpurun_out/instep_ab_$V.txt}
cd /tmp && export TMPDIR=/tmp
: > $OUT
for val in 1 0 1 0; do
  rm -rf /tmp/instep_$val
  env $V=$val rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/instep_$val -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-eval-forward --no-graph > /tmp/instep_$val.json 2> /dev/null
  echo "== $V=$val  ($(python3 -c "import json; d=json.loads(open('/tmp/instep_$val.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms/step eager under the profiler')"))" >> $OUT
  python3 - $val >> $OUT <<'PY'
import csv, glob, sys
f = glob.glob(f"/tmp/instep_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "wino" in r["Name"] or "ffm_" in r["Name"] or "cab_attn" in r["Name"]]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:9]:
    print(f'   {r["Name"].split("(")[0][-46:]:46s} calls {r["Calls"]:>4s}  avg {float(r["AverageNs"]) / 1e3:8.1f} us  min {float(r["MinNs"]) / 1e3:8.1f}  max {float(r["MaxNs"]) / 1e3:8.1f}  total/step {float(r["TotalDurationNs"]) / 1e3 / 15:8.1f} us')
PY
done
cat $OUT
