#!/usr/bin/env python3
"""Short per-kernel table from a rocprofv3 *kernel_stats.csv: python tools/kernel_stats.py <dir>"""
import csv, glob, re, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|")
    for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
        n = r["Name"]
        m = re.search(r"(radix_sort_onesweep_iteration|radix_sort_onesweep_global_offsets|transform_impl|scan_impl|lookback_scan|block_sort)", n)
        short = ("rocprim " + m.group(1) + (" <u64 key, u32 value>" if "unsigned long, unsigned int>" in n[:400] else " <u64 key>" if "unsigned long, rocprim" in n[:420] else " <u32 key, u32 value>" if "unsigned int, unsigned int>" in n[:400] else "")) if m else re.sub(r"\(.*", "", n).replace("void ", "")
        print("| `%s` | %s | %.1f | %.3f | %s |" % (short[:80], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Percentage"]))
