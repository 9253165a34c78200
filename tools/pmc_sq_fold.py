"""Fold rocprofv3 SQ counter passes (counter_collection CSVs; separate --pmc passes with --kernel-trace only) of
tools/conv_microbench.py into one line per kernel: the LARGEST dispatches of each kernel name (the 81 k / 259 k levels).
  python tools/pmc_sq_fold.py <out.txt> <csv> [<csv> ...]"""
import csv, re, sys, collections


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


acc = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> values of its biggest grid
grid_of = {}
for path in sys.argv[2:]:
    rows = [r for r in csv.DictReader(open(path)) if short(r["Kernel_Name"]).startswith(("k_spconv2", "k_spconv3", "k_wgrad_pairs"))]
    for r in rows:
        k = short(r["Kernel_Name"])
        g = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
        grid_of[k] = max(grid_of.get(k, 0), g)
    for r in rows:
        k = short(r["Kernel_Name"])
        g = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
        if g == grid_of[k]:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = []
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    out.append("%s  (grid %d threads, %d dispatches)" % (k, grid_of[k], len(next(iter(acc[k].values())))))
    wc = c.get("SQ_WAVE_CYCLES")
    for n in sorted(c):
        extra = ""
        if wc and n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS",
                        "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_VALU", "SQ_INST_CYCLES_VMEM"):
            extra = "  = %.1f %% of SQ_WAVE_CYCLES" % (100.0 * c[n] / wc)
        if n == "SQ_VALU_MFMA_BUSY_CYCLES" and c.get("SQ_BUSY_CYCLES"):
            extra = "  (SQ_BUSY_CYCLES %.4g)" % c["SQ_BUSY_CYCLES"]
        if n == "SQ_LDS_BANK_CONFLICT" and c.get("SQ_LDS_IDX_ACTIVE"):
            extra = "  = %.1f %% of SQ_LDS_IDX_ACTIVE" % (100.0 * c[n] / c["SQ_LDS_IDX_ACTIVE"])
        out.append("    %-28s %14.4g%s" % (n, c[n], extra))
open(sys.argv[1], "w").write("\n".join(out) + "\n")
print("\n".join(out))
