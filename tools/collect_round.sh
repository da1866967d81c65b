#!/bin/bash
# Everything the round's profiles/ are made of, on one GPU box:  bash tools/collect_round.sh r03
set -u
TAG=${1:-r03}
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
cd $GRAFT_REPO_ROOT
# config 5 first: behind the profiler passes below the same run measured 23.0 instead of 17.1 ms/step (profiles/README.md)
python tools/run_config5.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_config5.log; tail -n 1 $O/${TAG}_config5.log
python bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.log
tail -c 400 $O/${TAG}_bench_n1.json
export CABINET_FORCE_DDP=1
python tools/host_overhead.py 2>&1 | grep -v "Warn\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|socket.cpp" > $O/${TAG}_host_overhead.txt
cat $O/${TAG}_host_overhead.txt
# kernel timeline of the data-parallel step with RCCL forced at world size 1 (CABINET_FORCE_DDP exported above; the
# program itself directly behind `--`)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ddp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ddp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-roofline > $O/${TAG}_bench_ddp_world1_under_rocprof.json 2> /dev/null
cd $GRAFT_REPO_ROOT && python tools/ddp_overlap.py /tmp/prof_ddp gpurun_out/${TAG}_ddp_overlap_world1
unset CABINET_FORCE_DDP
cd /tmp
rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-graph > $O/${TAG}_bench_n1_under_rocprof.json 2> /dev/null
cd $GRAFT_REPO_ROOT && python tools/summarize_rocprof.py /tmp/prof_bench gpurun_out/${TAG}_bench_n1 "rocprofv3 --kernel-trace --stats of: python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-graph (eager enqueue, so that every kernel is a separate dispatch)"
bash tools/pmc_traffic.sh
bash tools/pmc_counters.sh $TAG
# the bench line again, now that the traffic profile of THIS build exists (bench.py checks the source digest)
cp $O/${TAG}_pmc_traffic.json $GRAFT_REPO_ROOT/profiles/${TAG}_pmc_traffic.json
cd $GRAFT_REPO_ROOT && python bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.log
tail -c 300 $O/${TAG}_bench_n1.json
