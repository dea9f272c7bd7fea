#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel (short names): python tools/pmc_summary.py <dir> [regex]"""
import csv, glob, re, sys, collections
d = sys.argv[1]; rx = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")[:60]
        if rx and not rx.search(name): continue
        acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[(name, row["Counter_Name"])] += 1
for name, cs in acc.items():
    print(name)
    for c, v in sorted(cs.items()):
        n = calls[(name, c)]
        print("   %-24s %16.0f total  %14.0f per dispatch (%d)" % (c, v, v / n, n))
