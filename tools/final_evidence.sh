#!/bin/bash
# What the round's profiles/ hold besides tools/collect_round.sh's artefacts:  bash tools/final_evidence.sh r06   (on the GPU box,
# AFTER collect_round.sh on the same box: the parity suite must not depend on what the box did before -- VERDICT r05 item 1b)
TAG=${1:-r06}
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
F='Warn\|amdgpu.ids\|return float\|Consider\|warn'
# ---- the GPU suite five times in a row on this (used) box
: > $O/${TAG}_gpu_tests_5x.log
for i in 1 2 3 4 5; do
  python -m pytest tests -m gpu -q 2>&1 | grep "passed\|failed\|FAILED\|error" | tail -3 | sed "s/^/run $i: /" >> $O/${TAG}_gpu_tests_5x.log
done
cat $O/${TAG}_gpu_tests_5x.log
python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/${TAG}_gpu_tests_full_run.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke" > $O/${TAG}_smoke.log; tail -1 $O/${TAG}_smoke.log
# ---- round 6 diagnostics: where the FFM's in-situ distance came from (flips, own-mask replay), with and without the double-precision
# re-decision; the step's run-to-run spread with MIOpen's default solvers; dependence on what the process did before
for c in 3 5; do
  python tools/diag_ffm_flips.py --config $c --out $O/${TAG}_ffm_flips_config$c.json > /dev/null 2>&1
  CABINET_FFM_EXACT_MASK=0 python tools/diag_ffm_flips.py --config $c --out $O/${TAG}_ffm_flips_config${c}_fp32_decisions.json > /dev/null 2>&1
done
python -c "
import json
for c in (3, 5):
    for v in ('', '_fp32_decisions'):
        d = json.load(open('gpurun_out/${TAG}_ffm_flips_config%d%s.json' % (c, v)))
        print('config', c, v or '(double re-decision)', 'flips', d['flips_own_mask'], 'dfsp vs fp64', '%.2e' % d['grads']['dfsp']['gpu_vs_f64_mask_replay'], 'own mask', '%.2e' % d['grads']['dfsp']['gpu_vs_own_mask_replay'])
"
# ---- K11: the 128-channel kernel against the 64-channel kernel and MIOpen, same box, same tensors
{ echo "== default (conv_out: 128 channels per workgroup, one wave per SIMD)"; python tools/time_conv3x3.py 2>&1 | grep -v "$F";
  echo "== CABINET_WINO_128=0 (round 5: 64 channels per workgroup)"; CABINET_WINO_128=0 python tools/time_conv3x3.py 2>&1 | grep "conv_out";
  echo "== CABINET_WINO_128=2 (persistent form, register epilogue)"; CABINET_WINO_128=2 python tools/time_conv3x3.py 2>&1 | grep "conv_out";
  echo "== config 5"; python tools/time_conv3x3.py --config5 --no-check 2>&1 | grep -v "$F";
  echo "== config 5, CABINET_WINO_128=0"; CABINET_WINO_128=0 python tools/time_conv3x3.py --config5 --no-check 2>&1 | grep "conv_out"; } > $O/${TAG}_conv3x3_ab.log
tail -30 $O/${TAG}_conv3x3_ab.log | cut -c1-170
# ---- the CAB's launch chain; small-GEMM core with k-contiguous operand images against round 3's
{ echo "== config 3 grid (8 x 256 x 32 x 32)"; python tools/time_cab_chain.py 2>&1 | grep -v "$F";
  echo "== CABINET_SG_KCONTIG=0 (round-3 small-GEMM core: two LDS dwords per MFMA)"; CABINET_SG_KCONTIG=0 python tools/time_cab_chain.py 2>&1 | grep "K6 \|project_out\|whole block";
  echo "== config 5 grid (2 x 256 x 64 x 32)"; python tools/time_cab_chain.py 2 64 32 2>&1 | grep -v "$F"; } > $O/${TAG}_cab_chain.txt
cat $O/${TAG}_cab_chain.txt | cut -c1-120
# ---- the data-parallel step at world size 1 with RCCL forced: both replay schedules
{ echo "== schedule chosen by the model (cabinet_amd/train.py::choose_ddp_schedule)"; CABINET_FORCE_DDP=1 python tools/ddp_segments.py 2>&1 | grep "^(\|^graphs\|^    its\|^host";
  echo "== CABINET_DDP_ONE_EVENT=0 (two events: round 5's default)"; CABINET_FORCE_DDP=1 CABINET_DDP_ONE_EVENT=0 python tools/ddp_segments.py 2>&1 | grep "^(\|^graphs";
  for v in 1 0; do echo "bench.py, RCCL forced at world 1, CABINET_DDP_ONE_EVENT=$v: $(CABINET_FORCE_DDP=1 CABINET_DDP_ONE_EVENT=$v python bench.py --no-cpu-baseline --no-kernel-roofline --no-eval-forward 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], "images/s", d["ms_per_step"], "ms/step", d["config"]["ddp_schedule"]["what"][:40])')"; done;
  echo "bench.py, single-GPU step (GraphedTrainStep): $(python bench.py --no-cpu-baseline --no-kernel-roofline --no-eval-forward 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], "images/s", d["ms_per_step"], "ms/step")')"; } > $O/${TAG}_ddp_segments.txt
cat $O/${TAG}_ddp_segments.txt | cut -c1-200
# ---- same-box A/B of the step: round-6 switches off one at a time
for sw in CABINET_WINO_128 CABINET_FFM_EXACT_MASK CABINET_SG_KCONTIG; do
  echo "$sw=0: $(env $sw=0 python bench.py --no-cpu-baseline --no-kernel-roofline --no-eval-forward 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], "images/s", d["ms_per_step"], "ms/step")')"
done > $O/${TAG}_step_ab.txt
echo "defaults: $(python bench.py --no-cpu-baseline --no-kernel-roofline --no-eval-forward 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], "images/s", d["ms_per_step"], "ms/step")')" >> $O/${TAG}_step_ab.txt
cat $O/${TAG}_step_ab.txt
