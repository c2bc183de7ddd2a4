#!/usr/bin/env python3
"""HBM bytes per launch and kernel class from the rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes of
profiles/collect.sh:
    traffic_summary.py <run dir with pmc2/ pmc3/ bench.json> > profiles/traffic.json
FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes, so reads
of the wide coalesced kind (here: 64-byte work items and 8-byte structure-of-arrays columns, one per lane)
are doubled; WRITE_SIZE is taken as reported (fp64 atomics are counted at more than their 8 bytes).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

run = sys.argv[1]


def per_class(pass_dir, counter):
    tot, calls = defaultdict(float), defaultdict(set)
    for path in glob.glob(os.path.join(run, pass_dir, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            cls = "k_trace" if "k_trace" in name else "k_shadow" if "k_shadow" in name else None
            if cls:
                tot[cls] += float(r["Counter_Value"])
                calls[cls].add(r["Dispatch_Id"])
    return {k: (tot[k], len(calls[k])) for k in tot}


fetch, write = per_class("pmc2", "FETCH_SIZE"), per_class("pmc3", "WRITE_SIZE")
workload = json.load(open(os.path.join(run, "bench.json")))["config"]["workload"]
out = {"workload": workload, "source": os.path.basename(os.path.normpath(run)), "kernels": {}}
for k in fetch:
    f_kib, n = fetch[k]
    w_kib, n_w = write[k]
    out["kernels"][k] = {
        "launches_profiled": n,
        "fetch_kib_reported": f_kib, "write_kib_reported": w_kib,
        "read_bytes_per_launch": 2 * f_kib * 1024 / n,
        "write_bytes_per_launch": w_kib * 1024 / n_w,
        "hbm_bytes_per_launch": 2 * f_kib * 1024 / n + w_kib * 1024 / n_w,
    }
json.dump(out, sys.stdout, indent=1)
print()
