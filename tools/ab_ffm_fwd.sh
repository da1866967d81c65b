cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ffm.py -x -q 2>&1 | tail -3
python bench.py --kernels-only --kernel-iters 30 2>&1 | grep "ffm_up_fwd \|ffm_up_bwd"
bash tools/kstats.sh ffm_up_fwd 20 2>&1 | grep "ffm_fwd_z\|ffm_pool\|ffm_gate"
bash tools/pmc_traffic.sh r05c ffm_up_fwd > /tmp/t.log 2>&1; tail -2 /tmp/t.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05c_pmc_traffic.json'))
pk=d.get('per_kernel_KiB',{}).get('ffm_up_fwd',{})
for c in ('FETCH_SIZE','WRITE_SIZE'):
    print(c, {k.split('::')[-1]: round(v/1024,1) for k,v in pk.get(c,{}).items() if abs(v)>1000})
PY
