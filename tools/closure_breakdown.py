"""Reads a rocprofv3 kernel trace of the closed-loop drive (tools/prof_mapper_cpp.sh with LOOP=1) and prints, for every loop-closure
refinement (from its first k_ov_keys to the last k_o3d_fold that follows), the kernel time by kernel.  Usage: closure_breakdown.py <kernel_trace.csv>"""
import collections, csv, re, sys
tr = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in tr)


def short(n):
    if "rocprim" in n:
        if "wrapped_" in n:
            return "rp:" + n.split("wrapped_")[1].split("<")[0]
        return "rp:init_lookback" if "init_lookback" in n else "rp:other"
    n = n.replace("(anonymous namespace)::", "")
    return re.sub(r"[(<].*$", "", n).split("::")[-1][:40]


folds = [e for e in ev if "k_o3d_fold" in e[2]]
keys = [e for e in ev if "k_ov_keys" in e[2]]
starts = [k for i, k in enumerate(keys) if i == 0 or k[0] - keys[i - 1][0] > 1_000_000]   # the two k_ov_keys of one call are back to back
for st in starts:
    nxt = min([s[0] for s in starts if s[0] > st[0]] + [st[0] + 40_000_000])
    t0, t1 = st[0], max(f[1] for f in folds if st[0] < f[0] < nxt)
    agg, cnt = collections.Counter(), collections.Counter()
    for s, e, n in ev:
        if t0 <= s <= t1 and ("o3d" in n or "k_ov" in n or "rocprim" in n or "k_vox" in n or "k_cell" in n or "k_src" in n or "k_gather" in n or "rocclr" in n or "k_heads" in n or "k_mask" in n or "k_compact" in n or "k_scan_total" in n):
            agg[short(n)] += e - s
            cnt[short(n)] += 1
    print(f"refinement: window {(t1 - t0) / 1e6:.2f} ms, kernels {sum(agg.values()) / 1e6:.2f} ms: " + ", ".join(f"{k} {v / 1e3:.0f} us x{cnt[k]}" for k, v in agg.most_common(12)))
