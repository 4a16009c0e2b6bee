"""Reads a rocprofv3 --kernel-trace CSV of tools/mapping_loop.py and reports, per scan of the steady state, how much of the
wall time the GPU was busy and where the host sat between kernels: gaps above a threshold, grouped by (kernel before ->
kernel after).  Usage: python tools/loop_gaps.py <kernel_trace.csv> [gap_us=4]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*$", "", n)
    n = re.sub(r"<.*$", "", n)
    return n.split("::")[-1][:40]
# steady state: the second half of the trace
ev = ev[len(ev) // 2:]
busy = sum(e - s for s, e, _ in ev)
wall = ev[-1][1] - ev[0][0]
gaps = collections.Counter(); gapt = collections.Counter()
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    g = (s1 - e0) / 1e3
    if g >= thr:
        k = (short(n0), short(n1)); gaps[k] += 1; gapt[k] += g
print(f"kernels {len(ev)}  wall {wall/1e6:.2f} ms  busy {busy/1e6:.2f} ms ({100*busy/wall:.1f} %)")
tot = sum(gapt.values())
print(f"gaps >= {thr} us: {sum(gaps.values())}, {tot/1e3:.2f} ms ({100*tot*1e3/wall:.1f} % of wall)")
for k, t in gapt.most_common(40):
    print(f"{gaps[k]:6d} x {t/gaps[k]:7.1f} us  = {t/1e3:7.2f} ms   {k[0]} -> {k[1]}")
