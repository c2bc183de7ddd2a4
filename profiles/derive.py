#!/usr/bin/env python3
"""Derived figures per kernel from the counter passes of profiles/collect.sh:
    derive.py <run dir with pmc1/ pmc6/>  > profiles/<name>_derived.json
  wait_fraction        SQ_WAIT_ANY / SQ_WAVE_CYCLES        share of a wave's life parked on s_waitcnt
  issue_stall_fraction SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  active_fraction      SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  valu_per_wave        SQ_INSTS_VALU / SQ_WAVES             vector instructions a wave executes
  lanes_per_valu       SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU   active lanes per vector instruction (64 = no divergence)
  fp64_share           (ADD + MUL + FMA + TRANS)_F64 / SQ_INSTS_VALU
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

run = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float))
for path in glob.glob(os.path.join(run, "pmc*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "rocclr" in name:
            continue
        tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
out = {}
for k, c in tot.items():
    g = lambda n: c.get(n, 0.0)  # noqa: E731
    wc, nv = g("SQ_WAVE_CYCLES"), g("SQ_INSTS_VALU")
    out[k] = {
        "wait_fraction": g("SQ_WAIT_ANY") / wc if wc else None,
        "issue_stall_fraction": g("SQ_WAIT_INST_ANY") / wc if wc else None,
        "active_fraction": g("SQ_ACTIVE_INST_ANY") / wc if wc else None,
        "valu_per_wave": nv / g("SQ_WAVES") if g("SQ_WAVES") else None,
        "lanes_per_valu": g("SQ_THREAD_CYCLES_VALU") / nv if nv and g("SQ_THREAD_CYCLES_VALU") else None,
        "fp64_share": (g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_FMA_F64")
                       + g("SQ_INSTS_VALU_TRANS_F64")) / nv if nv else None,
        "scalar_cache_hit_rate": g("SQC_DCACHE_HITS") / g("SQC_DCACHE_REQ") if g("SQC_DCACHE_REQ") else None,
    }
json.dump(out, sys.stdout, indent=1)
print()
