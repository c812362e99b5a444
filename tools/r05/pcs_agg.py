#!/usr/bin/env python3
"""Aggregate rocprofv3 PC-sampling CSVs under a directory: counts per (instruction, comment, other categorical columns)."""
import csv, sys, os, collections, glob
root = sys.argv[1]
for f in glob.glob(os.path.join(root, "**", "*pc_sampling*.csv"), recursive=True):
    agg = collections.Counter()
    with open(f, newline="") as fh:
        rd = csv.reader(fh)
        hdr = next(rd)
        drop = {i for i, h in enumerate(hdr) if h.lower() in ("sample_timestamp", "exec_mask", "dispatch_id", "correlation_id", "timestamp", "wave_id", "chiplet", "hw_id", "workgroup_id", "workgroup_id_x", "workgroup_id_y", "workgroup_id_z", "wave_in_group", "thread_id")}
        keep = [i for i in range(len(hdr)) if i not in drop]
        n = 0
        for row in rd:
            agg[tuple(row[i] for i in keep)] += 1; n += 1
    out = f[:-4] + ".agg.tsv"
    with open(out, "w") as o:
        o.write("count\t" + "\t".join(hdr[i] for i in keep) + "\n")
        for k, c in agg.most_common():
            o.write(str(c) + "\t" + "\t".join(k) + "\n")
    print(f, "samples", n, "distinct", len(agg), "header", hdr)
    if os.path.getsize(f) > 8 << 20:
        os.system(f"head -n 3000 '{f}' > '{f}.head' && rm '{f}'")
