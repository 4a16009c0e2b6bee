#!/bin/bash
# round 5: o3s_submap_insert_processed with its completion pending against the insert that waits (hooks build: O3S_INSERT_EAGER=1) on the
# compiled per-scan loop, ICP reference renewed every sweep and every 2 s.  -> gpurun_out/r05_lazy_<eager>_<pinned>.json
cd ${GRAFT_REPO_ROOT:-/root/repo}
for pin in 0 1; do
for e in 1 0; do
  if [ $e = 1 ]; then export O3S_INSERT_EAGER=1; else unset O3S_INSERT_EAGER; fi
  O3S_LIB_VARIANT=hooks PINNED=$pin SCANS=300 PREFETCH=2 PRELOAD=1 ALSO_REF_PERIOD=2.0 timeout -k 10 300 python3 tools/mapper_cpp_bench.py > gpurun_out/r05_lazy_${e}_${pin}.json 2> gpurun_out/r05_lazy_${e}_${pin}.err || exit 1
done
done
unset O3S_INSERT_EAGER
O3S_LIB_VARIANT=hooks SCANS=120 PREFETCH=2 PRELOAD=1 tools/prof_mapper_cpp.sh r05lazy
python3 tools/loop_gaps.py gpurun_out/prof_r05lazy/r05lazy_kernel_trace.csv > gpurun_out/r05_lazy_gaps.txt
echo done
