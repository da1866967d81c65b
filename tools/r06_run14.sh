cd $GRAFT_REPO_ROOT
# the GPU suite with its files in REVERSE order (order independence of the parity verdict: VERDICT r05 item 1b)
python -m pytest $(ls -r tests/test_*.py) -m gpu -q 2>&1 | grep "passed\|failed\|FAILED" | tail -3 | sed "s/^/files in reverse order: /" > gpurun_out/r06_gpu_tests_reverse_order.log
cat gpurun_out/r06_gpu_tests_reverse_order.log
