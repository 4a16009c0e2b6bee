#!/bin/bash
# round 5: the per-scan loop as THREE stages (one thread stages the raw sweep, a second pre-processes it, the mapping thread registers and
# inserts) against the two-stage driver, pageable and page-locked sweeps, reference renewed every sweep and every 2 s.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for pin in 0 1; do
for pf in 2 3; do
  PINNED=$pin SCANS=300 PREFETCH=$pf PRELOAD=1 ALSO_REF_PERIOD=2.0 timeout -k 10 300 python3 tools/mapper_cpp_bench.py > gpurun_out/r05_stage_${pf}_${pin}.json 2> gpurun_out/r05_stage_${pf}_${pin}.err || exit 1
done
done
echo done
