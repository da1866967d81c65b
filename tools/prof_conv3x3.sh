#!/bin/bash
# K11 on the GPU box: per-kernel durations (rocprofv3 kernel stats) and issue counters of the conv3x3 bench groups.
# usage: bash tools/prof_conv3x3.sh <tag> [group ...]      -> gpurun_out/<tag>_conv3x3_kstats.txt, gpurun_out/<tag>_conv3x3_pmc_counters.json
TAG=${1:-r05}; shift || true
GROUPS_RUN=${*:-"conv3x3_conva_fwd conv3x3_conva_bwd conv3x3_b1_fwd conv3x3_b1_bwd conv3x3_out_fwd conv3x3_out_bwd"}
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
: > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_conv3x3_kstats.txt
for G in $GROUPS_RUN; do
  echo "== $G" >> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_conv3x3_kstats.txt
  bash $GRAFT_REPO_ROOT/tools/kstats.sh $G 20 >> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_conv3x3_kstats.txt 2>&1
done
if [ -z "$NO_PMC" ]; then
  bash $GRAFT_REPO_ROOT/tools/pmc_counters.sh ${TAG}_conv3x3 $GROUPS_RUN > /tmp/pmc_conv.log 2>&1
fi
cat $GRAFT_REPO_ROOT/gpurun_out/${TAG}_conv3x3_kstats.txt
