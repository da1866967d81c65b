cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06l; mkdir -p $O
python -c "from cabinet_amd import build; print('fresh', build.is_fresh())"
python -m pytest tests/test_gpu_conv3x3.py tests/test_gpu_bn_cls.py tests/test_gpu_bn_act.py tests/test_gpu_insitu.py -x -q 2>&1 | tail -3
bash tools/instep_cycles2.sh | tail -1 | cut -c1-700
python tools/time_conv3x3.py 2>&1 | grep " fwd " | cut -c1-120
