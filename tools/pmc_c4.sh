#!/bin/bash
# usage (GPU box): tools/pmc_c4.sh  -> FETCH_SIZE / WRITE_SIZE of k_match at C4 (500k-pt scan vs 20M-pt map), separate passes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $ctr --kernel-include-regex k_match --output-format csv -d $R/gpurun_out/pmc_c4_$ctr -o p -- python3 $R/bench.py --steps 2 --warmup 1 --timing-only --scan 500000 --map 20000000 --voxel 0.02 > $R/gpurun_out/pmc_c4_$ctr.log 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob('$R/gpurun_out/pmc_c4_$ctr/*counter_collection.csv')
v = [float(r['Counter_Value']) for r in csv.DictReader(open(f[0])) if r['Counter_Name'] == '$ctr']
print('$ctr', 'avg/dispatch KB =', sum(v) / len(v), 'n =', len(v))
PY
done
