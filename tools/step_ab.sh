#!/bin/bash
# The graphed step with a switch at its default / off, alternating, 200 timed steps each (un-profiled):  bash tools/step_ab.sh <ENVVAR> <out>
V=${1:-CABINET_WINO_128}; OUT=${2:-$GRAFT_REPO_ROOT/gpurun_out/step_ab_$V.txt}
cd $GRAFT_REPO_ROOT
: > $OUT
for rep in 1 2 3 4; do
  for val in default 0; do
    if [ $val = default ]; then unset $V; else export $V=$val; fi
    python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-roofline --no-eval-forward 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$val', d['ms_per_step'], 'ms/step', d['value'], 'images/s;  fwd+loss+bwd only:', d['fwd_loss_bwd_only']['ms_per_step'], 'ms')" >> $OUT
    unset $V
  done
done
python - $OUT <<'PY'
import sys, statistics
rows = [l.split() for l in open(sys.argv[1]) if l.strip()]
by = {}
for r in rows:
    by.setdefault(r[0], []).append((float(r[1]), float(r[-2])))
with open(sys.argv[1], "a") as f:
    for k, v in by.items():
        f.write(f"{k}: median {statistics.median(x[0] for x in v):.3f} ms/step (fwd+loss+bwd only {statistics.median(x[1] for x in v):.3f}) over {len(v)} runs of 200 steps\n")
PY
cat $OUT
