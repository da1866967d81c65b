cd $GRAFT_REPO_ROOT; O=gpurun_out/r06b; mkdir -p $O
python tools/diag_ffm_flips.py --config 3 --out $O/ffm_flips_config3.json > $O/ffm_flips3.log 2>&1
python tools/diag_ffm_flips.py --config 5 --out $O/ffm_flips_config5.json > $O/ffm_flips5.log 2>&1
python -m pytest tests/test_gpu_ffm.py tests/test_gpu_model.py tests/test_gpu_insitu.py -x -q 2>&1 | tail -15 > $O/tests_subset.log
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/tests_full.log
python bench.py --kernels-only 2>&1 | grep -v Warn > $O/kernels.log
tail -3 $O/tests_subset.log; tail -3 $O/tests_full.log; grep ffm $O/kernels.log; grep flips $O/ffm_flips3.log
