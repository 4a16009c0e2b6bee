"""Reads a rocprofv3 --kernel-trace CSV of the per-scan loop (tools/mapping_loop.py or the compiled driver) and reports, per scan of the steady state, how much of the
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
kernel_time = sum(e - s for s, e, _ in ev)
wall = ev[-1][1] - ev[0][0]
# busy = the UNION of the kernel intervals (two streams overlap when a second thread pre-processes the next sweep); a gap is a
# stretch with no kernel running at all, attributed to (the kernel that ended last before it -> the kernel that ends it)
gaps = collections.Counter(); gapt = collections.Counter()
busy = 0
cur_s, cur_e, cur_n = ev[0]
for s1, e1, n1 in ev[1:]:
    if s1 > cur_e:
        busy += cur_e - cur_s
        g = (s1 - cur_e) / 1e3
        if g >= thr:
            k = (short(cur_n), short(n1)); gaps[k] += 1; gapt[k] += g
        cur_s, cur_e, cur_n = s1, e1, n1
    elif e1 > cur_e:
        cur_e, cur_n = e1, n1
busy += cur_e - cur_s
print(f"kernels {len(ev)}  wall {wall/1e6:.2f} ms  busy {busy/1e6:.2f} ms ({100*busy/wall:.1f} %)  sum of kernel durations {kernel_time/1e6:.2f} ms ({100*kernel_time/wall:.1f} % of wall)")
tot = sum(gapt.values())
print(f"gaps >= {thr} us: {sum(gaps.values())}, {tot/1e3:.2f} ms ({100*tot*1e3/wall:.1f} % of wall)")
for k, t in gapt.most_common(40):
    print(f"{gaps[k]:6d} x {t/gaps[k]:7.1f} us  = {t/1e3:7.2f} ms   {k[0]} -> {k[1]}")
