cd $GRAFT_REPO_ROOT; O=gpurun_out/r06h; mkdir -p $O
python -c "from cabinet_amd import build; print('fresh', build.is_fresh())"
python -m pytest tests/test_gpu_conv3x3.py -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke" | cut -c1-300
python tools/time_conv3x3.py 2>&1 | grep "wgrad\|worst" | cut -c1-150
python tools/time_conv3x3.py --config5 --no-check 2>&1 | grep "wgrad" | cut -c1-150
bash tools/instep_ab.sh CABINET_WINO_128 $O/instep_ab_wino128.txt | cut -c1-200
