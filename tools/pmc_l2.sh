#!/bin/bash
# L2 (TCC) hit / miss / request counts per kernel of the named bench.py kernel groups -- "HBM- or MFMA-bound at large HW?"
# (BASELINE config 5).  Counters only (no tracing domain), one rocprofv3 process per group.
# usage (on the GPU box):  bash tools/pmc_l2.sh <tag> <group ...>   ->  gpurun_out/<tag>_pmc_l2.json
set -u
TAG=${1:-r04}; shift || true
cd /tmp && export TMPDIR=/tmp
OUT=/tmp/pmc_l2
rm -rf $OUT; mkdir -p $OUT
for G in "$@"; do
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/$G -- python3 $GRAFT_REPO_ROOT/tools/run_kernels.py 4 $G > $OUT/$G.log 2>&1 || echo "pass failed: $G"
done
cd $GRAFT_REPO_ROOT && python3 - $OUT gpurun_out/${TAG}_pmc_l2.json <<'PY'
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.getcwd())
from cabinet_amd import build
root, out_path = sys.argv[1], sys.argv[2]
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
    ks = {}
    for k, cs in acc.items():
        if max(len(v) for v in cs.values()) < 4:
            continue  # operand set-up of an earlier group
        row = {c: round(sum(v) / len(v), 1) for c, v in cs.items()}
        hit, miss = row.get("TCC_HIT_sum", 0.0), row.get("TCC_MISS_sum", 0.0)
        if hit + miss:
            row["l2_hit_rate"] = round(hit / (hit + miss), 4)
        ks[k] = row
        print(f"{g:16s} {k[:60]:60s} L2 hit rate {row.get('l2_hit_rate', 0):.3f}  requests {row.get('TCC_REQ_sum', 0):.0f}")
    groups[g] = ks
json.dump({"source_digest": build.source_digest(), "batch": int(os.environ.get("CAB_B", "8")),
           "height": int(os.environ.get("CAB_H", "1024")), "width": int(os.environ.get("CAB_W", "1024")),
           "classes": int(os.environ.get("CAB_CLASSES", "8")),
           "method": "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum over tools/run_kernels.py 4 <group>; per-launch averages",
           "groups": groups}, open(out_path, "w"), indent=1, sort_keys=True)
print("wrote", out_path)
PY
