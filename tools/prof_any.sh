#!/bin/bash
# usage (GPU box): tools/prof_any.sh <tag> <bench args...>  -> gpurun_out/prof_<tag>/<tag>_kernel_stats.csv + a printed summary (rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $tag -- python3 $R/bench.py "$@" > $R/gpurun_out/prof_$tag.log 2>&1
echo "exit=$?" >> $R/gpurun_out/prof_$tag.log
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$R/gpurun_out/prof_$tag/${tag}_kernel_stats.csv')))
print('== $tag')
for r in rows[:14]:
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} min={float(r['MinNs'])/1e3:8.2f} max={float(r['MaxNs'])/1e3:9.2f} pct={r['Percentage']}")
PY
