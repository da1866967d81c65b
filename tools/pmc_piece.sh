#!/bin/bash
# counters of ONE piece's kernels: bash tools/pmc_piece.sh <piece> "<counters>"   (counters only, no tracing domain)
# <piece> is a piece of tools/trace_piece.py, or "group:<name>" for a bench.py kernel group (tools/run_kernels.py)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcp
case "$1" in
  group:*) rocprofv3 --pmc $2 --output-format csv -d /tmp/pmcp -- python3 $GRAFT_REPO_ROOT/tools/run_kernels.py 4 ${1#group:} > /tmp/pmcp.log 2>&1 ;;
  *) rocprofv3 --pmc $2 --output-format csv -d /tmp/pmcp -- python3 $GRAFT_REPO_ROOT/tools/trace_piece.py $1 4 > /tmp/pmcp.log 2>&1 ;;
esac
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmcp/**/*counter_collection.csv', recursive=True)
if not f:
    print(open('/tmp/pmcp.log').read()[-1500:]); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'][:70]
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); 
    cnt[(k, r['Counter_Name'])] += 1
for k, d in acc.items():
    if 'cabinet' not in k: continue
    print(k)
    for c, v in d.items(): print(f"    {c:32s} {v / cnt[(k, c)]:14.1f} per launch")
PY
