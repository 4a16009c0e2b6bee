#!/bin/bash
# usage (GPU box): tools/ab.sh "VAR=1 VAR2=x" "..."  -> bench.py --timing-only under each environment, three times each
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for envs in "$@"; do
  for rep in 1 2 3; do
    v=$(env $envs python3 bench.py --steps 20 --warmup 5 --timing-only ${AB_ARGS} 2>/dev/null | grep -o '"value": [0-9.]*')
    echo "[$envs] $v"
  done
done
