#!/bin/bash
# kernel timeline of the data-parallel step with RCCL forced at world size 1: bash tools/ddp_trace.sh <tag>
TAG=${1:-r05}
export CABINET_FORCE_DDP=1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ddp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ddp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-roofline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench_ddp_world1_under_rocprof.json 2> /dev/null
cd $GRAFT_REPO_ROOT && python tools/ddp_overlap.py /tmp/prof_ddp gpurun_out/${TAG}_ddp_overlap_world1
head -30 gpurun_out/${TAG}_ddp_overlap_world1.md
