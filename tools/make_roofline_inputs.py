"""Turns the rocprofv3 summaries of tools/r02_collect.sh (gpurun_out/r02z/) into the small JSON files bench.py reads for its
`roofline` object: the kernel-trace average of the dominant kernel over every in-chain launch, and the HBM bytes per launch
from the separate FETCH_SIZE / WRITE_SIZE passes (FETCH_SIZE x 2: the gfx950 128-byte-request correction of
MI355X_MICROARCH.md; WRITE_SIZE as is; both in KB per dispatch).  Usage: python tools/make_roofline_inputs.py <dir> <round>"""
import csv, json, os, re, sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r03z"
rnd = sys.argv[2] if len(sys.argv) > 2 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(root, "profiles", rnd)
os.makedirs(out_dir, exist_ok=True)
for tag in ("c2", "c4"):
    ks = os.path.join(src, f"kernel_stats_{tag}.csv")
    pm = os.path.join(src, f"pmc_traffic_{tag}.txt")
    if not (os.path.exists(ks) and os.path.exists(pm)):
        continue
    rows = [r for r in csv.DictReader(open(ks)) if "k_match2" in r["Name"]]
    calls = sum(int(r["Calls"]) for r in rows)
    avg_us = sum(float(r["TotalDurationNs"]) for r in rows) / calls / 1e3
    txt = open(pm).read()
    fetch = float(re.search(r"FETCH_SIZE\s+avg/dispatch=\s*([0-9.]+)", txt).group(1))
    write = float(re.search(r"WRITE_SIZE\s+avg/dispatch=\s*([0-9.]+)", txt).group(1))
    per_kernel = {}
    for kname in ("k_match2", "k_classify", "k_sel_ne", "k_sel_partial", "k_sel_finish", "k_normal_eq", "k_solve"):
        kr = [r for r in csv.DictReader(open(ks)) if re.search(r"\b" + kname + r"\b", r["Name"].replace("::", " ").replace("<", " "))]
        if kr:
            per_kernel[kname] = round(sum(float(r["TotalDurationNs"]) for r in kr) / sum(int(r["Calls"]) for r in kr) / 1e3, 3)
    rec = {"kernel": "k_match2", "k_match_avg_us": round(avg_us, 3), "launches": calls, "kernels_avg_us": per_kernel,
           "fetch_size_kb_per_dispatch": fetch, "write_size_kb_per_dispatch": write,
           "hbm_bytes_per_launch": int((2.0 * fetch + write) * 1024),
           "source": f"profiles/{rnd}/z_kernel_stats_{tag}.csv (rocprofv3 --kernel-trace --stats, every in-chain launch) and "
                     f"profiles/{rnd}/z_pmc_traffic_{tag}.txt (rocprofv3 --pmc FETCH_SIZE x2 [gfx950 correction] + --pmc WRITE_SIZE, separate passes)"}
    with open(os.path.join(out_dir, f"roofline_inputs_{tag}.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(tag, rec)
