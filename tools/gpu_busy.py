"""Dev tool: from a rocprofv3 --kernel-trace CSV, how busy was the GPU? Prints, for the second half of the trace (steady state),
the wall span, the union of all kernel intervals (time with at least one kernel resident), the sum of kernel durations and
the same split by kernel-name prefix given on the command line. Usage: python tools/gpu_busy.py kernel_trace.csv [prefix ...]"""
import csv, sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
t_mid = rows[len(rows) // 2][0]
rows = [r for r in rows if r[0] >= t_mid]
span = rows[-1][1] - rows[0][0]
union, cur_s, cur_e = 0, rows[0][0], rows[0][1]
for s, e, _ in rows[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
tot = sum(e - s for s, e, _ in rows)
print('kernels %d  span %.2f ms  busy(union) %.2f ms = %.1f %%  sum of durations %.2f ms' % (len(rows), span / 1e6, union / 1e6, 100.0 * union / span, tot / 1e6))
for pre in sys.argv[2:]:
    sel = [(s, e) for s, e, n in rows if pre in n]
    print('  %-20s %5d launches  %.2f ms (%.1f %% of span)' % (pre, len(sel), sum(e - s for s, e in sel) / 1e6, 100.0 * sum(e - s for s, e in sel) / span))
