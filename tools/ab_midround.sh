cd $GRAFT_REPO_ROOT
./tools/bin/buf_oob_probe
for v in "CABINET_WINO_KFAST=0" "CABINET_WINO_KFAST=1" "CABINET_WINO_PRIO=1"; do echo "== $v"; env $v python tools/time_conv3x3.py --no-check 2>&1 | grep "fwd\|dgrad" | cut -c1-75; done
echo "== default"; python tools/time_conv3x3.py --no-check 2>&1 | grep "fwd\|dgrad\|wgrad" | cut -c1-75
python -m pytest tests/test_gpu_bench_smoke.py tests/test_gpu_ohem.py tests/test_gpu_ddp_single.py tests/test_gpu_ddp_two_rank.py -q 2>&1 | tail -4
CABINET_FORCE_DDP=1 python tools/host_overhead.py 2>&1 | grep -v "Warn\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|socket.cpp" | tail -8
CABINET_FORCE_DDP=1 CABINET_DDP_INLINE_REDUCE=1 python tools/host_overhead.py 2>&1 | grep "RCCL forced: Graphed"
