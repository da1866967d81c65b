#!/bin/bash
# What the round's profiles/ hold besides tools/collect_round.sh's artefacts:  bash tools/final_evidence.sh r05   (on the GPU box)
TAG=${1:-r05}
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
F='Warn\|amdgpu.ids\|return float\|Consider\|warn'
python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/${TAG}_gpu_tests_full_run.log
tail -3 $O/${TAG}_gpu_tests_full_run.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke" > $O/${TAG}_smoke.log; tail -1 $O/${TAG}_smoke.log
(python tools/time_conv3x3.py; python tools/time_conv3x3.py --config5 --no-check) 2>&1 | grep -v "$F" > $O/${TAG}_conv3x3_ab.log
tail -22 $O/${TAG}_conv3x3_ab.log | cut -c1-170
# the CAB's launch chain piece by piece, with the round-4 forms beside the round-5 ones (same box, same tensors)
{ echo "== config 3 grid (8 x 256 x 32 x 32), round-5 defaults"; python tools/time_cab_chain.py 2>&1 | grep -v "$F";
  echo "== config 3 grid, CABINET_QKV_STATS_FUSED=0 (round 4's statistics launch)"; CABINET_QKV_STATS_FUSED=0 python tools/time_cab_chain.py 2>&1 | grep "K6 forward\|whole block";
  echo "== config 5 grid (2 x 256 x 64 x 32)"; python tools/time_cab_chain.py 2 64 32 2>&1 | grep -v "$F"; } > $O/${TAG}_cab_chain.txt
cat $O/${TAG}_cab_chain.txt | cut -c1-120
# the classifier tails: K12 against K7 + the stock 1x1 convolution (torch.profiler device times, standalone with a statistics pass)
{ echo "== K12 (default)"; python tools/cls_tail_probe.py 2>&1 | grep -v "$F";
  echo "== CABINET_BN_CLS=0: K7 + stock 1x1 convolution (round 4)"; CABINET_BN_CLS=0 python tools/cls_tail_probe.py 2>&1 | grep -v "$F"; } > $O/${TAG}_cls_tail_probe.txt
grep "total\|==" $O/${TAG}_cls_tail_probe.txt
CABINET_FORCE_DDP=1 python tools/ddp_segments.py 2>&1 | grep "^(\|^graphs\|^    its\|^host" > $O/${TAG}_ddp_segments.txt
CABINET_FORCE_DDP=1 CABINET_DDP_INLINE_REDUCE=1 python tools/ddp_segments.py 2>&1 | grep "^(d" | sed 's/^(d)/(d, round-4 order: CABINET_DDP_INLINE_REDUCE=1)/' >> $O/${TAG}_ddp_segments.txt
python tools/stream_overlap_probe.py 2>&1 | grep -v "$F\|capture_end" >> $O/${TAG}_ddp_segments.txt
cat $O/${TAG}_ddp_segments.txt | cut -c1-200
# same-box A/B of the step: round-5 operators off one at a time
for sw in CABINET_BN_CLS CABINET_CONV3X3 CABINET_ATTN_PROJ_FUSED; do
  echo "$sw=0: $(env $sw=0 python bench.py --no-cpu-baseline --no-kernel-roofline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], "images/s", d["ms_per_step"], "ms/step")')"
done > $O/${TAG}_step_ab.txt
python bench.py --height 2048 --width 1024 --batch 2 --classes 19 > $O/${TAG}_config5_bench_n1.json 2> $O/${TAG}_config5_bench_n1.log
python bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.log
echo "defaults: $(python -c 'import json; d=json.load(open("gpurun_out/'${TAG}'_bench_n1.json")); print(d["value"], "images/s", d["ms_per_step"], "ms/step")')" >> $O/${TAG}_step_ab.txt
cat $O/${TAG}_step_ab.txt
tail -c 400 $O/${TAG}_bench_n1.json
