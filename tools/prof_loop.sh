#!/bin/bash
# usage (on the GPU box): tools/prof_loop.sh <tag>   -> gpurun_out/prof_<tag>/<tag>_kernel_trace.csv of tools/mapping_loop.py (env passes through)
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $tag -- python3 $R/tools/mapping_loop.py > $R/gpurun_out/prof_$tag.log 2>&1
echo "exit=$?" >> $R/gpurun_out/prof_$tag.log
