#!/usr/bin/env python3
"""Markdown rows of CHANGELOG.md 5.2 from a directory of bench lines (tools/round_artifacts.sh): tools/bench_table.py profiles/r03"""
import json, os, sys
d = sys.argv[1]
def sci(v):
    e = len("%d" % v) - 1
    return "%.2f·10^%d" % (v / 10 ** e, e)
for w in ("chr1", "k63", "ecoli", "chr1_repeats", "chr1_dups"):
    p = os.path.join(d, "bench_%s.json" % w)
    if not os.path.exists(p):
        continue
    b = json.loads(open(p).read().strip().splitlines()[-1]); r = b["roofline"]; parts = r["kernel_ms_parts"]; c = b.get("cpu_baseline", {}); t = b.get("step_with_text", {})
    hb = ("%.1f GB = %.2f TB/s = %.2f of peak" % (r["traffic"] / 1e9, r["hbm_measured_gbps"] / 1e3, r["hbm_measured_frac"])) if r.get("traffic") else "–"
    print("| %s | %s | %.2f = %.2f / %.2f / %.2f | %.1f → %.2f TB/s = %.2f | %s | %.1f ms (text %.1f) = %s | %s / %s (%s) |" % (
        w, sci(b["value"]), b["ms_per_step"], parts["ingest_prefill"], parts["probe_prepass"], parts["search"], r["algorithmic_bytes_per_kmer"], r["achieved"] / 1e3, r["frac"], hb,
        t.get("ms", 0), t.get("ms_text", 0), sci(t.get("kmers_per_s", 1)), sci(c.get("search_only_value", 1)), sci(c.get("value", 1)), sci(b.get("cpu_baseline_16core", {}).get("value", 1)) if b.get("cpu_baseline_16core") else "-"))
    print("   stages:", {k: "%.2f B/k-mer in %.2f ms = %.2f" % (v["algorithmic_bytes_per_kmer"], v["ms"], v["frac"]) for k, v in r["stages"].items()})
    lc = r.get("lazy_counters_per_kmer", {}); rd = lc.get("reads", 0) or 1
    print("   per read:", {k: round(lc[k] / rd, 2) for k in ("table_entries", "probe_lines", "prepass_entries", "prepass_lines", "prepass_ktab", "seed_lookups", "text_anchors", "ktab_lookups", "deferred_strands", "deferred_slots", "text_windows", "chunks_probe", "chunks_search") if k in lc})
