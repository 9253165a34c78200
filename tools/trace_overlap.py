"""Dev tool: GPU timeline of ONE training step from a rocprofv3 --kernel-trace CSV: busy / idle / overlapped time, busy
time per queue, and the largest idle gaps with the kernels around them. Steps are delimited by k_adam launches.
  python tools/trace_overlap.py <kernel_trace.csv> [step_index_from_end=2]"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
ev.sort()
adam = [i for i, e in enumerate(ev) if e[2].startswith("k_adam")]
lo, hi = adam[-back - 1] + 1, adam[-back] + 1
step = ev[lo:hi]
t0, t1 = step[0][0], max(e[1] for e in step)
print("kernels in step: %d, span %.3f ms" % (len(step), (t1 - t0) / 1e6))
# sweep
pts = []
for s, e, *_ in step:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy = over = 0; depth = 0; last = t0
for t, d in pts:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    depth += d; last = t
print("busy %.3f ms, idle %.3f ms, >=2 kernels in flight %.3f ms, sum of kernel durations %.3f ms" %
      (busy / 1e6, (t1 - t0 - busy) / 1e6, over / 1e6, sum(e - s for s, e, *_ in step) / 1e6))
perq = defaultdict(lambda: [0, 0])
for s, e, n, q, st in step:
    perq[(q, st)][0] += e - s; perq[(q, st)][1] += 1
for k, (d, c) in sorted(perq.items(), key=lambda kv: -kv[1][0]):
    print("  queue %s stream %s: %4d kernels, %.3f ms busy" % (k[0], k[1], c, d / 1e6))
# idle gaps
gaps = []
cur_end = step[0][1]; prev = step[0]
for e in step[1:]:
    if e[0] > cur_end:
        gaps.append((e[0] - cur_end, prev[2][:50], e[2][:50]))
    if e[1] > cur_end:
        cur_end = e[1]; prev = e
gaps.sort(reverse=True)
print("idle gaps: %d, total %.3f ms; > 5 us: %d (%.3f ms)" % (len(gaps), sum(g[0] for g in gaps) / 1e6,
      sum(1 for g in gaps if g[0] > 5000), sum(g[0] for g in gaps if g[0] > 5000) / 1e6))
for g in gaps[:15]:
    print("  %7.1f us  after %-50s before %s" % (g[0] / 1e3, g[1], g[2]))
# short kernels
short = [e for e in step if e[1] - e[0] < 10000]
print("kernels < 10 us: %d, total %.3f ms" % (len(short), sum(e[1] - e[0] for e in short) / 1e6))
# timeline in 20 buckets: busy fraction and dominant kernel
nb = 24; w = (t1 - t0) / nb
for b in range(nb):
    a, z = t0 + b * w, t0 + (b + 1) * w
    acc = defaultdict(int); tot = 0
    for s, e, n, q, st in step:
        o = min(e, z) - max(s, a)
        if o > 0: acc[n[:40] + "|q" + str(q)] += o; tot += o
    top = sorted(acc.items(), key=lambda kv: -kv[1])[:2]
    print("  [%5.2f ms] load %.2f  %s" % (b * w / 1e6, tot / w, "; ".join("%s %.0f%%" % (k, 100 * v / w) for k, v in top)))
