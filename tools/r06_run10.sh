cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06j; mkdir -p $O
python -c "from cabinet_amd import build; print('fresh', build.is_fresh())"
bash tools/instep_ab.sh CABINET_WINO_128 $O/instep_ab_wino128.txt "wino_conv" | cut -c1-200
bash tools/instep_ab.sh CABINET_FFM_EXACT_MASK $O/instep_ab_ffm_exact.txt "ffm_pool|ffm_gate" | cut -c1-200
