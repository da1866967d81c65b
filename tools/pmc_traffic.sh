#!/bin/bash
# HBM traffic per launch of every bench.py kernel group: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes
# (counters only: no tracing domain beside them), each at two iteration counts so that operand set-up cancels.
# usage (on the GPU box):  bash tools/pmc_traffic.sh <tag> [group ...]   ->  gpurun_out/<tag>_pmc_traffic.json
# workload shape from the environment (tools/run_kernels.py): CAB_B CAB_H CAB_W CAB_CLASSES, default = BASELINE config 3
set -u
TAG=${1:-r06}; shift || true
cd /tmp && export TMPDIR=/tmp
GROUPS_ALL="cab_attn_fwd cab_attn_fwd_bf16x6 cab_attn_fwd_bf16x3 cab_attn_bwd ffm_up_fwd ffm_up_fwd_bf16x6 ffm_up_fwd_bf16x3 ffm_up_bwd bn_act_fwd bn_act_bwd bn_dwconv_fwd bn_dwconv_bwd stem_conv_fwd stem_conv_wrw pwconv_fwd pwconv_bwd ohem_up_pair_fwd ohem_up_pair_bwd cab_local_fwd cab_local_bwd cab_qkv_fwd cab_qkv_bwd conv3x3_conva_fwd conv3x3_conva_bwd conv3x3_b1_fwd conv3x3_b1_bwd conv3x3_out_fwd conv3x3_out_bwd cab_attn_proj_fwd bn_cls_out_fwd bn_cls_out_bwd bn_cls_head_fwd bn_cls_head_bwd"
GROUPS_RUN=${*:-$GROUPS_ALL}
OUT=/tmp/pmc_traffic
rm -rf $OUT; mkdir -p $OUT
for G in $GROUPS_RUN; do
  for C in FETCH_SIZE WRITE_SIZE; do
    for N in 3 6; do
      timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/$G/$C/$N -- python3 $GRAFT_REPO_ROOT/tools/run_kernels.py $N $G > $OUT/$G.$C.$N.log 2>&1 || echo "pass failed: $G $C $N"
    done
  done
done
cd $GRAFT_REPO_ROOT && python tools/summarize_traffic.py $OUT gpurun_out/${TAG}_pmc_traffic.json
