"""Per loop closure of a closed-loop run's kernel trace (tools/prof_mapper_cpp.sh with LOOP=1): wall time from the first overlap-key
kernel to the last fold of the refinement, and the kernels inside.  python3 tools/r04_closure_windows.py <kernel_trace.csv> [top]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::(o3s_cloud::)?", "", n)
    m = re.search(r"(radix_sort_onesweep_iteration|radix_sort_onesweep_global_offsets|merge_sort_block_merge|radix_sort_block_sort|scan_impl|init_lookback|merge_sort_block_sort|k_o3d_search_far|[a-z_0-9]+<\d>|k_[a-z_0-9]+|__amd_rocclr_\w+)", n)
    return m.group(1) if m else n[:30]


starts = [i for i, r in enumerate(rows) if "k_ov_keys" in r["Kernel_Name"]][::2]
for w, st in enumerate(starts):
    lim = starts[w + 1] if w + 1 < len(starts) else len(rows)
    end = max([j for j in range(st, lim) if "k_o3d_fold" in rows[j]["Kernel_Name"]] or [st])
    t0, t1 = int(rows[st]["Start_Timestamp"]), int(rows[end]["End_Timestamp"])
    agg = {}
    for j in range(st, end + 1):
        a = agg.setdefault(short(rows[j]["Kernel_Name"]), [0, 0.0])
        a[0] += 1
        a[1] += (int(rows[j]["End_Timestamp"]) - int(rows[j]["Start_Timestamp"])) / 1e3
    print(f"closure {w}: wall {(t1 - t0) / 1e3:.0f} us, {end - st + 1} kernels (all streams), kernel time {sum(v[1] for v in agg.values()):.0f} us")
    for n, (c, d) in sorted(agg.items(), key=lambda x: -x[1][1])[:top]:
        print(f"   {n:36s} {c:4d} {d:8.1f}")
