"""One sweep of the per-scan loop as the kernel trace shows it: every launch of a steady-state sweep in start order with its stream, duration and the
idle time in front of it (rocprofv3 --kernel-trace CSV of the compiled driver).  Usage: python tools/r05_sweep_timeline.py <kernel_trace.csv> [sweep_index_from_end=20]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 20
def short(n):
    n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*$", "", n)
    m = re.match(r"([^<]*)(<.*)?$", n)
    return (m.group(1).split("::")[-1] + ((m.group(2) or "")[:18]))[:48]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows)
# a sweep on the mapping stream starts with the patch mask of set_reference: k_mask directly followed (same queue) by the scan / k_compact_pm
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_ref_stats<") or e[2] == "k_ref_stats"]
if len(starts) < back + 2:
    print("too few sweeps in the trace"); sys.exit(1)
a, b = starts[-back - 1], starts[-back]
# walk back from the k_ref_stats to the k_mask that opens the sweep on the same queue
q = ev[a][3]
i = a
while i > 0 and not (ev[i][2].startswith("k_mask") and ev[i][3] == q):
    i -= 1
j = b
while j > 0 and not (ev[j][2].startswith("k_mask") and ev[j][3] == q):
    j -= 1
t0 = ev[i][0]
prev_end = {}
print(f"sweep from {ev[i][2]} to the next one's: {(ev[j][0] - t0) / 1e3:.1f} us")
for s, e, n, qq in ev[i:j]:
    gap = (s - prev_end[qq]) / 1e3 if qq in prev_end else 0.0
    prev_end[qq] = e
    print(f"{(s - t0) / 1e3:9.1f} us  q{qq:>3}  {(e - s) / 1e3:7.1f} us  (idle before {gap:6.1f})  {n}")
