cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06k; mkdir -p $O
python -c "from cabinet_amd import build; print('fresh', build.is_fresh())"
python -m pytest tests/test_gpu_ffm.py tests/test_gpu_insitu.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
for c in 3 5 6; do python tools/diag_ffm_flips.py --config $c --out $O/ffm_flips_config$c.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('$O/ffm_flips_config$c.json')); print('config', $c, 'flips', d['flips_own_mask'], 'dfsp', '%.2e' % d['grads']['dfsp']['gpu_vs_f64_mask_replay'], 'out', '%.2e' % d['grads']['out']['gpu_vs_own_mask_replay'])"; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke\] OK\|FFM ReLU"
bash tools/instep_ab.sh CABINET_FFM_EXACT_MASK $O/instep_ab_ffm_exact.txt "ffm_pool|ffm_gate|ffm_redecide" | cut -c1-200
bash tools/step_ab.sh CABINET_FFM_EXACT_MASK $O/step_ab_ffm_exact.txt | tail -3
bash tools/step_ab.sh CABINET_WINO_128 $O/step_ab_wino128.txt | tail -3
python bench.py --kernels-only 2>&1 | grep "ffm_up_fwd \|ffm_up_fwd_eval" | cut -c1-120
