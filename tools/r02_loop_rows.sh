#!/bin/bash
# the rows of DESIGN.md section 10 on the current tree -> gpurun_out/r02z/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r02z
mkdir -p $O
run() { out=$1; shift; env "$@" GEN_PROCS=12 timeout -k 10 500 python3 tools/mapping_loop.py > $O/$out.json 2> $O/$out.err; echo "$out: $(head -c 600 $O/$out.json)"; }
run loop60_normals NORMALS=1 CPU_SCANS=6
run loop60_far_prior NORMALS=1 PRED=0 CPU_SCANS=0
run loop60_raw NORMALS=0 CPU_SCANS=0
run loop60_raw_dense NORMALS=0 DENSE=1 CPU_SCANS=0
run lidar300_raw_dense_knn10 LIDAR=1 SCANS=300 STEP=0.25 NORMALS=0 DENSE=1 CPU_SCANS=0
run lidar300_raw_dense_knn30 LIDAR=1 SCANS=300 STEP=0.25 NORMALS=0 DENSE=1 KNN=30 KRAD=2.0 CPU_SCANS=0
run c5_2000_sweeps LIDAR=1 SCANS=2000 STEP=0.06 NORMALS=0 DENSE=1 KNN=30 KRAD=2.0 CPU_SCANS=0
