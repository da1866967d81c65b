cd $GRAFT_REPO_ROOT; O=gpurun_out/r06f; mkdir -p $O
python -c "from cabinet_amd import build; print('fresh', build.is_fresh())"
python -m pytest tests/test_gpu_conv3x3.py -x -q 2>&1 | tail -8 > $O/tests_conv3x3.log; tail -4 $O/tests_conv3x3.log
python tools/time_conv3x3.py 2>&1 | grep "conv_out\|worst" | cut -c1-150
CABINET_WINO_128=0 python tools/time_conv3x3.py 2>&1 | grep "conv_out" | cut -c1-150
python tools/time_conv3x3.py --config5 --no-check 2>&1 | grep "conv_out" | cut -c1-150
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/tests_full.log; tail -3 $O/tests_full.log
