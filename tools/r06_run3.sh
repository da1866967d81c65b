cd $GRAFT_REPO_ROOT; O=gpurun_out/r06c; mkdir -p $O
python -m pytest tests/test_gpu_conv3x3.py -x -q 2>&1 | tail -15 > $O/tests_conv3x3.log; tail -4 $O/tests_conv3x3.log
for pre in none nan junk small ffm; do python tools/diag_order_dependence.py --pre $pre --out $O/order_$pre.json > $O/order_$pre.log 2>&1; done
python - <<'PY'
import json
base=json.load(open('gpurun_out/r06c/order_none.json'))
for pre in ('nan','junk','small','ffm'):
    d=json.load(open(f'gpurun_out/r06c/order_{pre}.json'))
    diff=[k for k in base if k not in ('pre','second_run_differs_in','nan_grads') and base[k]!=d.get(k)]
    print(pre, 'differs from fresh process in', len(diff), 'entries', [k for k in diff if not k.startswith('grad.')][:12], 'nan:', d['nan_grads'][:3], '2nd run differs:', d['second_run_differs_in'][:5])
print('none: second run differs', base['second_run_differs_in'][:8])
PY
(python tools/time_conv3x3.py; CABINET_WINO_128=0 python tools/time_conv3x3.py) 2>&1 | grep -v "Warn\|amdgpu.ids" > $O/conv3x3_ab.log; cat $O/conv3x3_ab.log | cut -c1-180
