"""Per pass of every loop-closure refinement in a closed-loop kernel trace: durations (us) of keep / search / far / sums.
python3 tools/r04_closure_passes.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
line = []
for r in rows:
    n = r["Kernel_Name"]
    for key, tag in (("k_o3d_keep", "keep"), ("k_o3d_search_far", "far"), ("k_o3d_search<", "near"), ("k_o3d_corr", "sums"), ("k_ov_keys", "OV")):
        if key in n:
            d = round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1)
            if tag == "OV":
                if line:
                    print(" ".join(line))
                    line = []
            else:
                line.append(f"{tag}:{d}")
                if tag == "sums":
                    line.append("|")
            break
print(" ".join(line))
