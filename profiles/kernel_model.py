#!/usr/bin/env python3
"""Per-kernel figures that bench.py's roofline entry prices its live timings with, from the counter
passes of profiles/collect.sh (passes 1 2 3 5 6) on one bench configuration:
    kernel_model.py <run dir with bench.json pmc1/ pmc2/ pmc3/ pmc5/ pmc6/> [more run dirs]
merges one entry per workload into profiles/kernel_model.json:
  valu_per_unit        SQ_INSTS_VALU per ray (k_trace: primary + secondary) or per shadow segment (k_shadow)
  fp64_share           (ADD + MUL + FMA + TRANS)_F64 / SQ_INSTS_VALU
  hbm_bytes_per_unit   2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes; on gfx950 FETCH_SIZE tallies the 128-byte
                       requests of wide coalesced reads as 64 bytes, WRITE_SIZE is taken as reported)
  wait_fraction, active_fraction, lanes_per_valu   as in derive.py
Every frame of a profiled run is the same frame (same seed), so a run's counter totals divided by the
number of frames it rendered (= dispatches of k_resolve) are per-frame figures; the units per frame come
from the same configuration's bench.json.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "kernel_model.json")


def klass(name):
    """The class a launch is timed under by the engine: closest hit + shading (one kernel, or the
    k_closest_stream / k_shade pair on large scenes), shadow segments (k_shadow or k_shadow_stream)."""
    if "k_trace" in name or "k_closest_stream" in name or "k_shade<" in name or "k_shade(" in name:
        return "k_trace"
    return "k_shadow" if "k_shadow" in name else "k_resolve" if "k_resolve" in name else None


def one_pass(run, n):
    """counter -> class -> total, and the number of frames the pass rendered"""
    tot, frames = defaultdict(lambda: defaultdict(float)), set()
    for path in glob.glob(os.path.join(run, f"pmc{n}", "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            c = klass(r["Kernel_Name"])
            if c == "k_resolve":
                frames.add(r["Dispatch_Id"])
            if c:
                tot[r["Counter_Name"]][c] += float(r["Counter_Value"])
    return tot, max(len(frames), 1)


def main():
    model = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for run in sys.argv[1:]:
        bench = json.load(open(os.path.join(run, "bench.json")))
        steps = bench["steps"]
        per_sample = bench["rays_per_sample"], bench["shadow_segments_per_sample"]
        rays = bench["value"] * 1e6 * bench["ms_per_step"] * 1e-3  # primary + secondary rays per frame
        units = {"k_trace": rays, "k_shadow": rays / per_sample[0] * per_sample[1]}
        per_frame = defaultdict(dict)  # class -> counter -> per-frame value
        for n in (1, 2, 3, 5, 6):
            tot, frames = one_pass(run, n)
            for counter, by_class in tot.items():
                for c, v in by_class.items():
                    per_frame[c][counter] = v / frames
        entry = {"source": os.path.basename(os.path.normpath(run)), "steps_in_bench": steps, "kernels": {}}
        for c in ("k_trace", "k_shadow"):
            g = lambda k: per_frame[c].get(k, 0.0)  # noqa: E731
            nv, wc = g("SQ_INSTS_VALU"), g("SQ_WAVE_CYCLES")
            f64 = g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_TRANS_F64")
            entry["kernels"][c] = {
                "source": entry["source"],
                "unit": "ray" if c == "k_trace" else "shadow segment",
                "units_per_frame": units[c],
                "valu_per_unit": nv / units[c],
                "fp64_share": f64 / nv if nv else None,
                "hbm_bytes_per_unit": (2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0 / units[c],
                "read_bytes_per_unit": 2.0 * g("FETCH_SIZE") * 1024.0 / units[c],
                "write_bytes_per_unit": g("WRITE_SIZE") * 1024.0 / units[c],
                "wait_fraction": g("SQ_WAIT_ANY") / wc if wc else None,
                "active_fraction": g("SQ_ACTIVE_INST_ANY") / wc if wc else None,
                "lanes_per_valu": g("SQ_THREAD_CYCLES_VALU") / nv if nv and g("SQ_THREAD_CYCLES_VALU") else None,
            }
        model[bench["config"]["workload"]] = entry
    with open(OUT, "w") as f:
        json.dump(model, f, indent=1)
        f.write("\n")
    print("wrote", OUT, "workloads:", list(model))


if __name__ == "__main__":
    main()
