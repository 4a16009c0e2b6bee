#!/bin/bash
# round 5 experiment: HIP stream priorities for the receiving side (low) / the mapping side (high) on the compiled per-scan loop
# (hooks build: O3S_X_PRIO, csrc/cloud_dev.h make_stream).  Writes gpurun_out/r05_prio_<mode>_<pinned>.json
cd ${GRAFT_REPO_ROOT:-/root/repo}
for pin in 1 0; do
for m in 0 1 2 3; do
  O3S_LIB_VARIANT=hooks O3S_X_PRIO=$m PINNED=$pin SCANS=300 PREFETCH=2 PRELOAD=1 timeout -k 10 200 python3 tools/mapper_cpp_bench.py > gpurun_out/r05_prio_${m}_${pin}.json 2> gpurun_out/r05_prio_${m}_${pin}.err || exit 1
done
done
echo done
