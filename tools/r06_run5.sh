cd $GRAFT_REPO_ROOT; O=gpurun_out/r06e; mkdir -p $O
python tools/diag_ffm_flips.py --config 6 --out $O/ffm_flips_config6.json > $O/ffm_flips6.log 2>&1; python -c "
import json; d=json.load(open('$O/ffm_flips_config6.json')); print({k:v for k,v in d.items() if k!='grads'}); print({k:v for k,v in d['grads'].items() if k in ('dfsp','dlow','convblk.bn.bias')})"
python -m pytest tests/test_gpu_qkv.py tests/test_gpu_ffm.py tests/test_gpu_attn.py -x -q 2>&1 | tail -5
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/tests_full.log; tail -3 $O/tests_full.log
CABINET_SG_KCONTIG=0 python bench.py --kernels-only 2>&1 | grep "cab_qkv\|ffm_up\|cab_attn_bwd\|cab_attn_proj" | cut -c1-120
python bench.py --kernels-only 2>&1 | grep "cab_qkv\|ffm_up\|cab_attn_bwd\|cab_attn_proj" | cut -c1-120
python tools/time_cab_chain.py 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -30
