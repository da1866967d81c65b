cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-graph > $GRAFT_REPO_ROOT/gpurun_out/r05a_bench_n1_under_rocprof.json 2> /dev/null
cd $GRAFT_REPO_ROOT
python tools/summarize_rocprof.py /tmp/prof_bench gpurun_out/r05a_bench_n1 "config 3 (mid-round): rocprofv3 --kernel-trace --stats of: python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-graph"
bash tools/pmc_counters.sh r05b conv3x3_conva_fwd conv3x3_out_fwd conv3x3_out_bwd ohem_up_pair_fwd > /tmp/pmc.log 2>&1
bash tools/pmc_traffic.sh r05b ohem_up_pair_fwd conv3x3_out_fwd > /tmp/pmct.log 2>&1
tail -3 /tmp/pmct.log
