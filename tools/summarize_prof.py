#!/usr/bin/env python3
"""Condense rocprofv3 csv output (kernel stats + pmc passes) into a short text summary and the profiles/traffic.json entry of the
profiled workload.  HBM bytes per step = sum over the kernels of one step of (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch:
FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (it reports half of a wide read stream; calibrated for this
access pattern in profiles/r01_v2/fetch_calibration.txt); launches per step come from the kernel trace."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
print("# rocprofv3 summary for", os.path.basename(out))
for f in sorted(glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)):
    print("## kernel stats (--kernel-trace --stats):", os.path.relpath(f, out))
    for i, row in enumerate(csv.reader(open(f))):
        if i < 14:
            print(",".join(x[:70] for x in row))
per_launch = defaultdict(dict)
launches = {}
fills = defaultdict(list)   # every fill's own counter value: one-off fills (batch creation) must not be spread over the steps
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
            if "fillBuffer" in kn:
                fills[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for kn in sorted(sums):
        if not (kn.startswith("fin_") or "fillBuffer" in kn):
            continue
        for c in sorted(sums[kn]):
            n = cnt[kn][c]
            per_launch[kn][c] = sums[kn][c] / n
            launches[kn] = n
            print("pmc %-44s %-24s launches=%d  sum=%.6g  per_launch=%.6g" % (kn, c, n, sums[kn][c], sums[kn][c] / n))
# traffic entry: kernels of a step and how often each runs per step (bench ran steps + warmup = 3 steps)
try:
    bench = json.load(open(os.path.join(out, "trace.json")))
    steps = bench["steps"] + bench["warmup"]
    # (the profiled command runs more steps than it times: bench.py's step_with_text passes; the pre-pass kernel runs once per step)
    for kname, nl in launches.items():
        if kname.startswith(("fin_probe_kernel", "fin_pair_prepass_kernel", "fin_fast_prepass_kernel", "fin_probe_pair_kernel")):
            steps = nl
    step_kernels = [k for k in per_launch if k.startswith(("fin_pack", "fin_probe", "fin_pair_prepass", "fin_fast_prepass", "fin_search", "fin_route", "fin_stream", "fin_walk")) or "fillBuffer" in k]
    total = 0.0; parts = {}
    for k in step_kernels:
        if "FETCH_SIZE" in per_launch[k] and "WRITE_SIZE" in per_launch[k]:
            per_step = launches[k] / steps
            if "fillBuffer" in k:
                # only fills that recur with every step count (the (-1,-1) prefill when there is one): a value seen at least `steps`
                # times; a fill that happens once -- guard bytes and buffers at batch creation -- is not traffic of a step
                b = 0.0
                for cname, mul in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
                    seen = defaultdict(int)
                    for v in fills[cname]:
                        seen[round(v, -1)] += 1
                    b += mul * 1024 * sum(v * (n // steps) for v, n in seen.items() if n >= steps and v >= 1024)
                per_step = 1.0
            else:
                b = (2 * per_launch[k]["FETCH_SIZE"] + per_launch[k]["WRITE_SIZE"]) * 1024 * per_step
            parts[k] = int(b); total += b
    h = hashlib.sha256()
    d = os.path.join(ROOT, "finito_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")) or f == "fin_capi.cpp":
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    wl = bench["config"]["workload"].split(" = ")[0]
    entry = {"%s:%d:%s" % (wl, bench["config"]["reads_per_gpu"], bench["config"]["kernel"]): {
        "hbm_bytes_per_step": int(total), "parts": parts, "kernel_src_sha16": h.hexdigest()[:16],
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `bench.py %s` (tools/profile_gpu.sh): per step the sum over its kernels of (2*FETCH_SIZE + WRITE_SIZE)*1024" % " ".join(sys.argv[2:])}}
    json.dump(entry, open(os.path.join(out, "traffic_entry.json"), "w"), indent=1)
    print("## traffic entry:", json.dumps(entry))
except Exception as e:   # a missing pass must not lose the summary
    print("## traffic entry not made:", e)
