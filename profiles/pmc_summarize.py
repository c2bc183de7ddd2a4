#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter CSVs per kernel name.  usage: pmc_summarize.py <counter_collection.csv>..."""
import csv
import sys
from collections import defaultdict

for path in sys.argv[1:]:
    tot = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"][:48]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
    for k, c in tot.items():
        print(k, f"dispatches={len(calls[k])}", " ".join(f"{n}={v:.4g}" for n, v in sorted(c.items())))
