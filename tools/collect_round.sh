#!/bin/bash
# Everything the round's profiles/ are made of, on one GPU box:  bash tools/collect_round.sh r04
# Two workloads get the SAME set of artefacts:
#   BASELINE config 3 (8 x 3 x 1024 x 1024, 8 classes)   -> gpurun_out/<tag>_*
#   BASELINE config 5 (2 x 3 x 2048 x 1024, 19 classes)  -> gpurun_out/<tag>_config5_*
# per workload: the bench JSON line (roofline + cpu_baseline), rocprofv3 kernel stats of the eager run of the same bench,
# PMC HBM traffic and PMC issue counters of the section-8 kernel groups; config 3 also host overhead and the world-1 RCCL timeline.
set -u
TAG=${1:-r06}
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
C5="--height 2048 --width 1024 --batch 2 --classes 19"
S8_GROUPS="cab_attn_fwd cab_attn_fwd_bf16x6 cab_attn_fwd_bf16x3 cab_attn_bwd ffm_up_fwd ffm_up_bwd ohem_up_pair_fwd ohem_up_pair_bwd cab_local_fwd cab_local_bwd cab_qkv_fwd cab_qkv_bwd conv3x3_conva_fwd conv3x3_conva_bwd conv3x3_b1_fwd conv3x3_b1_bwd conv3x3_out_fwd conv3x3_out_bwd cab_attn_proj_fwd bn_cls_out_fwd bn_cls_out_bwd bn_cls_head_fwd bn_cls_head_bwd"
cd $GRAFT_REPO_ROOT
# ---- benches first, on the fresh box (behind the profiler passes MIOpen's find database has been seen to change solver choices)
python bench.py $C5 > $O/${TAG}_config5_bench_n1.json 2> $O/${TAG}_config5_bench_n1.log
tail -c 300 $O/${TAG}_config5_bench_n1.json
python bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.log
tail -c 300 $O/${TAG}_bench_n1.json
# the forward-only (eval) line of both workloads (round 6: VERDICT r05 item 7)
python bench.py --eval --no-kernel-roofline $C5 > $O/${TAG}_config5_bench_eval.json 2> /dev/null
python bench.py --eval > $O/${TAG}_bench_eval.json 2> $O/${TAG}_bench_eval.log
tail -c 300 $O/${TAG}_bench_eval.json
export CABINET_FORCE_DDP=1
python tools/host_overhead.py 2>&1 | grep -v "Warn\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|socket.cpp" > $O/${TAG}_host_overhead.txt
cat $O/${TAG}_host_overhead.txt
# kernel timeline of the data-parallel step with RCCL forced at world size 1 (the program itself directly behind `--`)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ddp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ddp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-roofline > $O/${TAG}_bench_ddp_world1_under_rocprof.json 2> /dev/null
cd $GRAFT_REPO_ROOT && python tools/ddp_overlap.py /tmp/prof_ddp gpurun_out/${TAG}_ddp_overlap_world1
unset CABINET_FORCE_DDP
# ---- rocprofv3 kernel stats of the eager run of the same bench, both workloads
cd /tmp
rm -rf /tmp/prof_bench /tmp/prof_bench5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-graph > $O/${TAG}_bench_n1_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench5 -- python3 $GRAFT_REPO_ROOT/bench.py $C5 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-graph > $O/${TAG}_config5_bench_n1_under_rocprof.json 2> /dev/null
cd $GRAFT_REPO_ROOT
python tools/summarize_rocprof.py /tmp/prof_bench gpurun_out/${TAG}_bench_n1 "config 3: rocprofv3 --kernel-trace --stats of: python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-graph (eager enqueue, so that every kernel is a separate dispatch)"
python tools/summarize_rocprof.py /tmp/prof_bench5 gpurun_out/${TAG}_config5_bench_n1 "config 5: rocprofv3 --kernel-trace --stats of: python3 bench.py $C5 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-graph"
# ---- PMC passes: counters only, one rocprofv3 process per kernel group and counter set
bash tools/pmc_traffic.sh $TAG $S8_GROUPS ffm_up_fwd_eval
bash tools/pmc_counters.sh $TAG $S8_GROUPS
export CAB_B=2 CAB_H=2048 CAB_W=1024 CAB_CLASSES=19
bash tools/pmc_traffic.sh ${TAG}_config5 $S8_GROUPS
bash tools/pmc_counters.sh ${TAG}_config5 $S8_GROUPS
bash tools/pmc_l2.sh ${TAG}_config5 cab_attn_fwd cab_attn_bwd
unset CAB_B CAB_H CAB_W CAB_CLASSES
bash tools/pmc_l2.sh ${TAG} cab_attn_fwd cab_attn_bwd
# ---- the bench lines again, now that the traffic profiles of THIS build exist (bench.py checks digest and shape)
cp $O/${TAG}_pmc_traffic.json $O/${TAG}_config5_pmc_traffic.json $GRAFT_REPO_ROOT/profiles/
cd $GRAFT_REPO_ROOT
python bench.py $C5 > $O/${TAG}_config5_bench_n1.json 2> $O/${TAG}_config5_bench_n1.log
tail -c 300 $O/${TAG}_config5_bench_n1.json
python bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.log
tail -c 300 $O/${TAG}_bench_n1.json
