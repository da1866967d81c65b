cd $GRAFT_REPO_ROOT; O=gpurun_out/r06a; mkdir -p $O
python tools/diag_ffm_flips.py --config 3 --out $O/ffm_flips_config3.json > $O/ffm_flips3.log 2>&1
python tools/diag_ffm_flips.py --config 5 --out $O/ffm_flips_config5.json > $O/ffm_flips5.log 2>&1
python tools/diag_step_determinism.py --out $O/det_plain.json > $O/det_plain.log 2>&1
python tools/diag_step_determinism.py --poison --out $O/det_poison.json > $O/det_poison.log 2>&1
python tools/diag_step_determinism.py --deterministic --out $O/det_det.json > $O/det_det.log 2>&1
python tools/diag_step_determinism.py --mode small --poison --out $O/det_small_poison.json > $O/det_small_poison.log 2>&1
python bench.py > $O/bench_n1.json 2> $O/bench_n1.log
tail -c 600 $O/ffm_flips3.log; tail -c 300 $O/det_plain.log
