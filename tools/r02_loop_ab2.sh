#!/bin/bash
# A/B of the per-scan loop over one environment knob: tools/r02_loop_ab2.sh VAR val1 val2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r02q
var=$1; shift
for v in "$@"; do
  export $var=$v
  LIDAR=1 SCANS=${SCANS:-150} STEP=0.25 NORMALS=1 GEN_PROCS=12 CPU_SCANS=0 timeout -k 10 300 python3 tools/mapping_loop.py > gpurun_out/r02q/loop_${var}_$v.json 2> gpurun_out/r02q/loop_${var}_$v.err
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/r02q/loop_${var}_$v.json'))
print('$var=$v', d['gpu_hz'], d['gpu_ms_per_scan_median'], d['stage_ms_median'], d['icp_iterations_median'], d['pose_error_m_max'])"
done
