cd $GRAFT_REPO_ROOT
for i in 3 4 5; do python -m pytest tests -m gpu -q 2>&1 | grep "passed\|failed" | tail -1 | sed "s/^/final tree, run $i (fresh box): /"; done > gpurun_out/r06_gpu_tests_final_tree_more.log
cat gpurun_out/r06_gpu_tests_final_tree_more.log
