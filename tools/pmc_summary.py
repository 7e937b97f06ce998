#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, per counter, mean value per dispatch."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            print(k, {a: round(sum(b) / len(b), 1) for a, b in v.items()}, "dispatches", max(len(b) for b in v.values()))
