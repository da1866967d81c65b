cd $GRAFT_REPO_ROOT
O=gpurun_out
python tools/time_conv3x3.py --config5 --no-check 2>&1 | grep "fwd\|dgrad\|wgrad\|cat" > $O/r05_conv3x3_ab_config5.log; cat $O/r05_conv3x3_ab_config5.log | cut -c1-170
CABINET_FORCE_DDP=1 python tools/ddp_segments.py 2>&1 | grep "^(\|^graphs\|^    its\|^host" > $O/r05_ddp_segments_core.txt
CABINET_FORCE_DDP=1 CABINET_DDP_INLINE_REDUCE=1 python tools/ddp_segments.py 2>&1 | grep "^(d" | sed 's/^(d)/(d, round-4 order: CABINET_DDP_INLINE_REDUCE=1)/' >> $O/r05_ddp_segments_core.txt
cat $O/r05_ddp_segments_core.txt
