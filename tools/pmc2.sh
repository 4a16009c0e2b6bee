#!/bin/bash
# usage (GPU box): tools/pmc2.sh <tag> <kernel-regex> "<counters pass 1>" ["<counters pass 2>" ...]   (counter passes only: no trace domains)
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
kre=$1; shift
cd /tmp && export TMPDIR=/tmp
p=0
for ctrs in "$@"; do
  p=$((p+1))
  timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-include-regex "$kre" --output-format csv -d $R/gpurun_out/pmc_${tag}_$p -o p -- python3 $R/bench.py --steps 2 --warmup 1 --timing-only ${PMC_ARGS} > $R/gpurun_out/pmc_${tag}_$p.log 2>&1
  python3 - <<PY
import csv, collections, glob
f = glob.glob('$R/gpurun_out/pmc_${tag}_$p/*counter_collection.csv')
if not f:
    print('no counter file'); raise SystemExit
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f[0])):
    k = (r['Kernel_Name'][:34], r['Counter_Name'])
    acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
for (kn, cn), (v, n) in sorted(acc.items()):
    print(f"{kn:34s} {cn:34s} avg/dispatch={v/n:14.1f} n={n}")
PY
done
