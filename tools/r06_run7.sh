cd $GRAFT_REPO_ROOT; O=gpurun_out/r06g; mkdir -p $O
python -c "from cabinet_amd import build; print('fresh', build.is_fresh())"
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/tests_full.log; tail -3 $O/tests_full.log
python tools/time_conv3x3.py 2>&1 | grep "conv_out\|worst" | cut -c1-150
for v in 0 1; do echo "CABINET_FFM_EXACT_MASK=$v"; CABINET_FFM_EXACT_MASK=$v python bench.py --kernels-only 2>&1 | grep "ffm_up_fwd \|ffm_up_fwd_eval" | cut -c1-120; done
