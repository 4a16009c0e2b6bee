#!/bin/bash
# GPU box: per-pass durations of the loop-closure search kernels on tools/r04_closure.py (hooks build; O3S_O3D_KDBG / O3S_O3D_G pass through)
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-t}
cd /tmp && export TMPDIR=/tmp
export O3S_LIB_VARIANT=hooks
REPS=1 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04z_o3d_$tag -o t -- python3 $R/tools/r04_closure.py > /dev/null 2>&1
cd $R
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/r04z_o3d_$tag/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
out={"near":[], "far":[], "sums":[]}
for r in rows:
    n=r["Kernel_Name"]
    if "k_o3d_search" in n or "k_o3d_corr" in n:
        out["far" if "search_far" in n else ("near" if "search" in n else "sums")].append(round((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,1))
print("$tag", out)
PY
