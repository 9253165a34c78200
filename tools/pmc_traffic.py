"""Fold two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) into per-kernel HBM bytes per
launch:  python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KB; on gfx950 FETCH_SIZE reports half of a wide
(16 B/lane) coalesced read stream (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is uncalibrated."""
import csv, json, re, sys, collections


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


def fold(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter and r["Kernel_Name"].lstrip("void ").startswith("k_"):
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


fetch, write = fold(sys.argv[1], "FETCH_SIZE"), fold(sys.argv[2], "WRITE_SIZE")
out = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace only) on `bench.py --steps 2 "
                "--warmup 1 --no-pipeline --no-cpu-baseline`; values are per-dispatch means in the counter's KB unit. bytes = "
                "(2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE reports half of a wide (16 B/lane) coalesced read "
                "stream (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is uncalibrated.",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f = sum(fetch.get(k, [0])) / max(1, len(fetch.get(k, [])))
    w = sum(write.get(k, [0])) / max(1, len(write.get(k, [])))
    out["kernels"][k] = {"dispatches": len(fetch.get(k, [])), "fetch_kb_mean": round(f, 1), "write_kb_mean": round(w, 1),
                         "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("kernels:", len(out["kernels"]))
