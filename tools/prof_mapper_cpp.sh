#!/bin/bash
# usage (on the GPU box): tools/prof_mapper_cpp.sh <tag>   -> gpurun_out/prof_<tag>/<tag>_kernel_trace.csv of the COMPILED per-scan loop
# (tests/cpp/mapper_loop.cpp).  tools/mapper_cpp_bench.py writes the scenario and builds the driver (env passes through: SCANS,
# PREFETCH, PRELOAD, ...); the driver itself then runs once more under rocprofv3 with the same switches.
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
D=/tmp/o3s_prof_$tag
KEEP_DIR=$D python3 $R/tools/mapper_cpp_bench.py > $R/gpurun_out/prof_$tag.bench.json 2> $R/gpurun_out/prof_$tag.bench.err || exit 1
[ -n "$PREFETCH" ] && [ "$PREFETCH" != "0" ] && export O3S_DRIVER_PREFETCH=$PREFETCH
[ "$PRELOAD" = "1" ] && export O3S_DRIVER_PRELOAD=1
[ "$PINNED" = "1" ] && export O3S_DRIVER_PINNED=1
[ "$LOOP" = "1" ] && export O3S_DRIVER_LOOP_CLOSURES=1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $tag -- $D/mapper_loop $D/scenario.bin $D/out_prof.txt $D/timing_prof.txt > $R/gpurun_out/prof_$tag.log 2>&1
echo "exit=$?" >> $R/gpurun_out/prof_$tag.log
cp $D/timing_prof.txt $R/gpurun_out/prof_$tag.timing.txt
