#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r02q
LIDAR=1 SCANS=60 STEP=0.25 NORMALS=1 GEN_PROCS=1 CPU_SCANS=0 tools/prof_loop.sh r02q
python3 tools/loop_gaps.py gpurun_out/prof_r02q/r02q_kernel_trace.csv 4 > gpurun_out/r02q/gaps.txt 2>&1
tail -5 gpurun_out/prof_r02q.log
cat gpurun_out/r02q/gaps.txt
