cd $GRAFT_REPO_ROOT; O=gpurun_out/r06d; mkdir -p $O
python -m pytest tests/test_gpu_conv3x3.py -x -q 2>&1 | tail -15 > $O/tests_conv3x3.log; tail -4 $O/tests_conv3x3.log
for pre in none nan ffm; do python tools/diag_order_dependence.py --pre $pre --out $O/order_$pre.json > $O/order_$pre.log 2>&1; done
python - <<'PY'
import json
base=json.load(open('gpurun_out/r06d/order_none.json'))
for pre in ('nan','ffm'):
    d=json.load(open(f'gpurun_out/r06d/order_{pre}.json'))
    diff=[k for k in base if k not in ('pre','second_run_differs_in','nan_grads') and base[k]!=d.get(k)]
    print(pre, 'differs from fresh process in', len(diff), 'entries', sorted(k for k in diff if k.startswith('cap.sb') or not k.startswith(('grad.','cap.')))[:12])
PY
python -m pytest tests/test_gpu_ffm.py tests/test_gpu_model.py tests/test_gpu_insitu.py -x -q 2>&1 | tail -15 > $O/tests_subset.log; tail -3 $O/tests_subset.log
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/tests_full.log; tail -3 $O/tests_full.log
python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.log; tail -3 $O/bench_n1.log; python -c "
import json; d=json.load(open('$O/bench_n1.json')); print(d['value'], d['ms_per_step'], d.get('eval_forward'), d.get('hot_path_ms_per_step'), d['library'])
for k in d['kernels']: print(k['kernel'][:40], k['ms_per_launch'], k['bound'], k['frac'], k.get('frac_mfma'), k.get('frac_hbm'), k.get('bytes_formula_suspect','')[:20])"
python bench.py --eval --no-kernel-roofline > $O/bench_eval.json 2> $O/bench_eval.log; tail -c 1500 $O/bench_eval.json
