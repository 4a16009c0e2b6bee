"""Summarises a rocprofv3 --kernel-trace csv: per kernel name count / avg us, and for the last K 'registrations' (runs of kernels
that start with k_read_prep) the span first-start -> last-end, the sum of kernel time and the idle time inside."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    k = r["Kernel_Name"].split("(")[0][:60]
    acc[k][0] += 1; acc[k][1] += d
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:60s} n={n:6d} avg={t/n:8.2f} us total={t*1e-3:9.3f} ms")
# registrations
regs, cur = [], []
for r in rows:
    if "k_read_prep" in r["Kernel_Name"] and cur:
        regs.append(cur); cur = []
    cur.append(r)
if cur: regs.append(cur)
regs = [g for g in regs if "k_read_prep" in g[0]["Kernel_Name"]][-30:]
spans, busy, nk = [], [], []
for g in regs:
    s, e = int(g[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in g)
    spans.append((e - s) * 1e-3); busy.append(sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in g)); nk.append(len(g))
import statistics as st
print(f"registrations analysed: {len(regs)}; kernels per registration {st.median(nk)}; span first-start..last-end median {st.median(spans):.1f} us; "
      f"sum of kernel time median {st.median(busy):.1f} us; idle inside {st.median(spans) - st.median(busy):.1f} us")
g = regs[-1]
prev = None
for r in g:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) * 1e-3 if prev else 0.0
    print(f"   +{gap:6.2f} gap  {(e - s) * 1e-3:7.2f} us  {r['Kernel_Name'].split('(')[0][:70]}")
    prev = e
