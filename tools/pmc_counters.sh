#!/bin/bash
# Issue / occupancy counters per kernel of every bench.py kernel group: rocprofv3 --pmc in SEPARATE passes (counters only: no
# tracing domain beside them; <= 4 SQ counters per pass, four passes), one rocprofv3 process per group so that kernels launched by several
# groups with different shapes (sg_gemm, gemm_kmajor ...) are not averaged across them.
# usage (on the GPU box):  bash tools/pmc_counters.sh <tag> [group ...]   ->  gpurun_out/<tag>_pmc_counters.json
set -u
TAG=${1:-r06}; shift || true
cd /tmp && export TMPDIR=/tmp
GROUPS_ALL="cab_attn_fwd cab_attn_fwd_bf16x6 cab_attn_fwd_bf16x3 cab_attn_bwd ffm_up_fwd ffm_up_bwd ohem_up_pair_fwd ohem_up_pair_bwd cab_local_fwd cab_local_bwd cab_qkv_fwd cab_qkv_bwd conv3x3_conva_fwd conv3x3_conva_bwd conv3x3_b1_fwd conv3x3_b1_bwd conv3x3_out_fwd conv3x3_out_bwd"
GROUPS_RUN=${*:-$GROUPS_ALL}
OUT=/tmp/pmc_counters
rm -rf $OUT; mkdir -p $OUT
for G in $GROUPS_RUN; do
  i=0
  for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/$G/p$i -- python3 $GRAFT_REPO_ROOT/tools/run_kernels.py 4 $G > $OUT/$G.p$i.log 2>&1 || echo "pass failed: $G p$i"
  done
done
cd $GRAFT_REPO_ROOT && python tools/summarize_counters.py $OUT gpurun_out/${TAG}_pmc_counters.json
