#!/bin/bash
# round 5: does the number of hardware queues the HIP runtime spreads its streams over (GPU_MAX_HW_QUEUES, default 4) decide how well the
# receiving thread's pre-processing overlaps the mapping thread's call?  -> gpurun_out/r05_hwq_<n>_<pinned>.json
cd ${GRAFT_REPO_ROOT:-/root/repo}
for pin in 1 0; do
for q in 0 2 8 16; do
  if [ $q = 0 ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  PINNED=$pin SCANS=300 PREFETCH=2 PRELOAD=1 ALSO_REF_PERIOD=2.0 timeout -k 10 300 python3 tools/mapper_cpp_bench.py > gpurun_out/r05_hwq_${q}_${pin}.json 2> gpurun_out/r05_hwq_${q}_${pin}.err || exit 1
done
done
echo done
