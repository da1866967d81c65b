# the driver's round-end sequence on a fresh box: GPU tests (-x), smoke(), the bench line with the driver's flags
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06_rehearsal; mkdir -p $O
python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 > $O/gpu_tests.log; tail -2 $O/gpu_tests.log | head -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke\]" | cut -c1-160 > $O/smoke.log; tail -1 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.log
python -c "
import json; d=json.load(open('$O/bench.json'))
print(d['metric'], d['value'], d['unit'], d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['traffic'], 'cpu', d['cpu_baseline']['value'], 'eval', d['eval_forward']['value'], 'hot', d['hot_path_ms_per_step']['value'], d['library'])
print([ (k['kernel'].split(' ')[0], k['traffic'] is not None) for k in d['kernels'] if k['traffic'] is None][:5])"
tail -3 $O/bench.log
