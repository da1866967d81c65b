#!/bin/bash
# What the round's profiles/ hold besides tools/collect_round.sh's artefacts:  bash tools/final_evidence.sh r05   (on the GPU box)
TAG=${1:-r05}
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/${TAG}_gpu_tests_full_run.log
tail -3 $O/${TAG}_gpu_tests_full_run.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke" > $O/${TAG}_smoke.log; tail -1 $O/${TAG}_smoke.log
(python tools/time_conv3x3.py; python tools/time_conv3x3.py --config5 --no-check) 2>&1 | grep -v "Warn\|amdgpu.ids\|return float\|Consider" > $O/${TAG}_conv3x3_ab.log
tail -22 $O/${TAG}_conv3x3_ab.log | cut -c1-170
CABINET_FORCE_DDP=1 python tools/ddp_segments.py 2>&1 | grep "^(\|^graphs\|^    its\|^host" > $O/${TAG}_ddp_segments.txt
CABINET_FORCE_DDP=1 CABINET_DDP_INLINE_REDUCE=1 python tools/ddp_segments.py 2>&1 | grep "^(d" | sed 's/^(d)/(d, round-4 order: CABINET_DDP_INLINE_REDUCE=1)/' >> $O/${TAG}_ddp_segments.txt
python tools/stream_overlap_probe.py 2>&1 | grep -v "Warn\|amdgpu.ids\|capture_end" >> $O/${TAG}_ddp_segments.txt
cat $O/${TAG}_ddp_segments.txt | cut -c1-200
python bench.py --height 2048 --width 1024 --batch 2 --classes 19 > $O/${TAG}_config5_bench_n1.json 2> $O/${TAG}_config5_bench_n1.log
python bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.log
tail -c 400 $O/${TAG}_bench_n1.json
