"""Dev tool: every kernel of ONE training step from a rocprofv3 --kernel-trace CSV, in start order, one line per launch:
start (us after the step's first kernel), duration, gap to the previous kernel of the same queue, queue, name.
Steps are delimited by k_adam launches.   python tools/trace_step_dump.py <kernel_trace.csv> [step_index_from_end=2]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
adam = [i for i, e in enumerate(ev) if e[2].startswith("k_adam")]
lo, hi = adam[-back - 1] + 1, adam[-back] + 1
step = ev[lo:hi]
t0 = step[0][0]
qs = {}
for q in sorted(set(e[3] for e in step)):
    qs[q] = len(qs)
last = {}
print("kernels: %d; non-irx (no k_ prefix): %d" % (len(step), sum(1 for e in step if not e[2].lstrip("void ").startswith("k_"))))
for s, e, n, q in step:
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    name = n.replace("void ", "")[:70]
    print("%9.1f %7.1f %7.1f  q%d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, qs[q], name))
