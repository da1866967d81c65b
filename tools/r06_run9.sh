cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06i; mkdir -p $O
python -c "from cabinet_amd import build; print('fresh', build.is_fresh())"
python -m pytest tests/test_gpu_ffm.py tests/test_gpu_insitu.py -x -q 2>&1 | tail -3
bash tools/instep_ab.sh CABINET_WINO_128 $O/instep_ab_wino128.txt | cut -c1-200
bash tools/instep_ab.sh CABINET_FFM_EXACT_MASK $O/instep_ab_ffm_exact.txt | grep "==\|ffm_pool\|ffm_gate\|ffm_fwd_z" | cut -c1-200
