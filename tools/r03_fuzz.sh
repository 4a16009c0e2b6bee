#!/bin/bash
# round 3: the differential campaign on the final tree (three new seeds, every case computed three times so that use_graph cases replay
# a captured graph) and the per-iteration repro records of the round-2 cases ADVICE lists -> gpurun_out/r03x/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r03x
mkdir -p $O
for seed in 41 42 43; do
  SEED=$seed CASES=700 REPEAT=3 timeout -k 10 600 python3 tools/fuzz_parity.py > $O/x_fuzz_parity_seed$seed.json 2> $O/fuzz_$seed.err
  tail -c 300 $O/x_fuzz_parity_seed$seed.json | head -c 300; echo
done
for sc in "31 192" "31 576" "32 212" "32 218" "33 642"; do
  set -- $sc
  SEED=$1 timeout -k 10 300 python3 tools/fuzz_repro.py $2 > $O/x_fuzz_seed$1_case$2_repro.txt 2>&1
done
echo done
