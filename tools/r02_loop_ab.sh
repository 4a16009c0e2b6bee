#!/bin/bash
# A/B of the per-scan loop (config 5's scan model): hinted side pipelines vs measured index ranges
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r02q
for v in hint nohint; do
  if [ $v = nohint ]; then export O3S_NO_HINT=1; else unset O3S_NO_HINT; fi
  LIDAR=1 SCANS=${SCANS:-150} STEP=0.25 NORMALS=1 GEN_PROCS=12 CPU_SCANS=0 timeout -k 10 300 python3 tools/mapping_loop.py > gpurun_out/r02q/loop_$v.json 2> gpurun_out/r02q/loop_$v.err
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/r02q/loop_$v.json'))
print('$v', d['gpu_hz'], d['gpu_ms_per_scan_median'], d['stage_ms_median'], d['pose_error_m_max'])"
done
