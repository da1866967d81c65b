cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cyc_all
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/cyc_all -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-eval-forward --no-graph > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/instep_vs_loop_cycles.py /tmp/cyc_all profiles/r06_pmc_counters.json > gpurun_out/r06_instep_vs_loop_cycles.txt; cat gpurun_out/r06_instep_vs_loop_cycles.txt
