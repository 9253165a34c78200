"""Dev tool: rocprofv3 kernel_stats.csv -> per-step time and launches by kernel family.
  python tools/stats_groups.py <kernel_stats.csv> <steps in the capture>"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
groups = [("conv fwd/dgrad (k_spconv*)", r"k_spconv[234]?<|k_updgrad"), ("conv wgrad (k_wgrad*, k_spconv2_wgrad)", r"k_wgrad3|k_wgrad_pairs|k_spconv2_wgrad|k_spconv_wgrad"),
          ("split reduces (k_wgrad_reduce, k_pairs_reduce)", r"k_wgrad_reduce|k_pairs_reduce"), ("stem", r"k_stem"),
          ("BatchNorm (k_bn_*)", r"k_bn_"), ("weight images (k_permute*)", r"k_permute"),
          ("kernel maps / hash / pyramid (k_kmap, k_voxel, k_fill, k_ds, k_down, k_pairs_count/write, k_tile)", r"k_kmap|k_voxel|k_fill_|k_ds_|k_down_|k_pairs_count|k_pairs_write|k_tile|k_hash|k_keys|k_coords|k_quantize|k_batch_off|k_bev"),
          ("radix sort (k_rs_*)", r"k_rs_"), ("head MLPs (k_mlp_*)", r"k_mlp_"), ("GRU (k_gru_*)", r"k_gru"),
          ("other irx (k_*)", r"(^|void )k_"), ("rocBLAS (Cijk)", r"Cijk"), ("memset / memcpy", r"__amd_rocclr"), ("ATen", r".")]
acc = {g[0]: [0.0, 0.0] for g in groups}
for r in rows:
    name = r["Name"].strip('"')
    for g, pat in groups:
        if re.search(pat, name):
            acc[g][0] += float(r["TotalDurationNs"]); acc[g][1] += float(r["Calls"]); break
tot = sum(v[0] for v in acc.values()); cal = sum(v[1] for v in acc.values())
print("per step over %g steps: %.3f ms of kernel time, %.0f launches" % (steps, tot / steps / 1e6, cal / steps))
for g, _ in groups:
    t, c = acc[g]
    if c: print("  %-100s %7.3f ms %6.1f launches" % (g, t / steps / 1e6, c / steps))
