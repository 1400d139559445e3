#!/usr/bin/env python3
"""Condense rocprofv3 csv output (kernel stats + pmc passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
print("# rocprofv3 summary for", os.path.basename(out))
for f in sorted(glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)):
    print("## kernel stats (--kernel-trace --stats):", os.path.relpath(f, out))
    for i, row in enumerate(csv.reader(open(f))):
        if i < 8:
            print(",".join(row))
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    sums = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            kn = row.get("Kernel_Name", "?").split("(")[0][:60]
            sums[kn][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[kn][row["Counter_Name"]] += 1
    for kn in sums:
        if "fin_search" not in kn and "fin_probe" not in kn:
            continue
        for c in sums[kn]:
            n = cnt[kn][c]
            print("pmc %-44s %-24s launches=%d  sum=%.6g  per_launch=%.6g" % (kn, c, n, sums[kn][c], sums[kn][c] / n))
