#!/bin/bash
# rocprofv3 PMC passes (counters only, no tracing domains) for the hand-written kernels + calibration.
# usage (on the GPU box): bash tools/pmc_passes.sh <tag>
set -u
TAG=${1:-r01}
# the calibration / probe binaries are not tracked: (re)build them for gfx950 when missing
for t in pmc_calib mfma_probe; do
  [ -x $GRAFT_REPO_ROOT/tools/$t ] || hipcc --offload-arch=gfx950 -O3 $GRAFT_REPO_ROOT/tools/$t.hip -o $GRAFT_REPO_ROOT/tools/$t
done
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/k$i -- python3 $GRAFT_REPO_ROOT/tools/run_kernels.py 5 > $OUT/k$i.log 2>&1
done
timeout 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- $GRAFT_REPO_ROOT/tools/pmc_calib > $OUT/calib1.log 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- $GRAFT_REPO_ROOT/tools/pmc_calib > $OUT/calib2.log 2>&1
find $OUT -name "*.csv" | head -20
