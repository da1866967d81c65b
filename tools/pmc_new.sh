#!/bin/bash
# HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE, counters only) for the kernel groups added after the first PMC
# run, one group per rocprofv3 process so that per-kernel averages are not mixed across shapes.
# usage (on the GPU box): bash tools/pmc_new.sh   -> gpurun_out/pmc_<group>.json
set -u
cd /tmp && export TMPDIR=/tmp
for G in bn_act dwconv ohem cab ffm_up; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$G
  rm -rf $OUT; mkdir -p $OUT
  i=0
  for C in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 400 rocprofv3 --pmc $C --output-format csv -d $OUT/k$i -- python3 $GRAFT_REPO_ROOT/tools/run_kernels.py 3 $G > $OUT/k$i.log 2>&1
  done
  rm -f $GRAFT_REPO_ROOT/gpurun_out/pmc_$G.json
  (cd $GRAFT_REPO_ROOT && python tools/summarize_pmc.py $OUT gpurun_out/pmc_$G.json | tail -1)
  rm -rf $OUT
done
