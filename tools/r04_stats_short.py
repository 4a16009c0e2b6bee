"""Prints a rocprofv3 kernel_stats.csv with short kernel names: python3 tools/r04_stats_short.py <csv> [divide_calls_by]"""
import csv
import re
import sys

div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    short = re.sub(r"\(anonymous namespace\)::(o3s_cloud::)?", "", n)
    m = re.search(r"(radix_sort_onesweep_iteration|radix_sort_onesweep_global_offsets|merge_sort_block_merge|radix_sort_block_sort|scan_impl|init_lookback|merge_sort_block_sort|[a-z_0-9]+<\d>|k_[a-z_0-9]+|__amd_rocclr_\w+)", short)
    key = m.group(1) if m else short[:40]
    kt = "u64,u32" if "unsigned long, unsigned int>" in n[:400] else ("u64,-" if "empty_type" in n[:500] else "")
    t = float(r["TotalDurationNs"]) / 1e3 / div
    tot += t
    print(f"{key:38s} {kt:8s} calls {int(r['Calls']) / div:8.1f}  avg {float(r['AverageNs']) / 1e3:8.1f} us  total {t:9.1f} us")
print("sum", round(tot, 1), "us")
