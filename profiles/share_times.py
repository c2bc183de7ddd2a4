#!/usr/bin/env python3
"""What each rank of an N-GPU run has to render, timed on ONE GPU: for world = 1, 2, 4, 8 every rank's
share of the frame (mirt_render_tiles with that rank and world, 16x16 tiles) is rendered in turn, a few
frames each, one frame at a time and with two frames in flight.  The slowest rank's time is what an
N-GPU frame takes before its gather; world-1 time / that is the speed-up the tile split allows.
Not a scaling measurement: no second GPU, no gather.

    python profiles/share_times.py [--config cfg2|cfg3] [--frames 8] > profiles/r3_share_times_<cfg>.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # as bench.py: kernel arguments in device memory

PRESETS = {
    "cfg2": dict(scene="Test5", width=1920, height=1080, spp=16, over=dict(max_ray_depth=5)),
    "cfg3": dict(scene="Test5", width=1920, height=1080, spp=16,
                 over=dict(max_ray_depth=8, glossy_samples=8, shadow_samples=16)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=sorted(PRESETS), default="cfg2")
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--in-flight", type=int, default=2, choices=(2, 3, 4))
    args = ap.parse_args()
    import torch
    from raytracer_amd import DeviceScene, HostScene, default_config, tile_buffer_pixels
    p = PRESETS[args.config]
    hs = HostScene(os.path.join(ROOT, "fixtures", "ASCII", f"{p['scene']}.txt"))
    ds = DeviceScene(hs, 0)
    cam, _ = hs.camera(0)
    cfg = default_config(**p["over"])
    W, H, spp, tile = p["width"], p["height"], p["spp"], 16
    dev = torch.device("cuda", 0)
    depth = args.in_flight
    streams = [torch.cuda.Stream(dev) for _ in range(depth)]
    out = {"config": args.config, "workload": f"{p['scene']} {W}x{H} {spp}spp {p['over']}", "tile": tile,
           "frames_per_rank": args.frames, "frames_in_flight": depth, "worlds": {}}
    for world in (1, 2, 4, 8):
        ranks = []
        for rank in range(world):
            n = tile_buffer_pixels(W, H, tile, tile, rank, world)
            bufs = [torch.empty(n * 3, dtype=torch.float64, device=dev) for _ in range(depth)]
            torch.cuda.synchronize(dev)

            def begin(slot):
                ds.render_tiles_begin(cam, cfg, W, H, spp, tile, tile, rank, world, bufs[slot].data_ptr(),
                                      streams[slot].cuda_stream, slot)

            for slot in range(depth):  # sizes the working memory
                begin(slot)
                st = ds.render_tiles_end(slot)
            alone = pipelined = float("inf")
            for _ in range(2):  # the better of two runs each: the first run after a change of share can catch the clocks ramping
                t0 = time.perf_counter()
                for _ in range(args.frames):
                    begin(0)
                    st = ds.render_tiles_end(0)
                alone = min(alone, (time.perf_counter() - t0) / args.frames * 1e3)
                t0 = time.perf_counter()
                for i in range(args.frames + depth):
                    if i >= depth:
                        ds.render_tiles_end(i % depth)
                    if i < args.frames:
                        begin(i % depth)
                pipelined = min(pipelined, (time.perf_counter() - t0) / args.frames * 1e3)
            ranks.append({"rank": rank, "ms_one_at_a_time": alone, "ms_two_in_flight": pipelined,
                          "device_ms": st["device_ms"], "rays": st["primary_rays"] + st["secondary_rays"],
                          "shadow_segments": st["shadow_segments"], "launches": st["launches"]})
        worst = max(r["ms_two_in_flight"] for r in ranks)
        worst_alone = max(r["ms_one_at_a_time"] for r in ranks)
        work = [r["rays"] + r["shadow_segments"] for r in ranks]
        out["worlds"][str(world)] = {"ranks": ranks, "max_ms_two_in_flight": worst, "max_ms_one_at_a_time": worst_alone,
                                     "work_max_over_mean": max(work) / (sum(work) / len(work))}
    one = out["worlds"]["1"]
    for world, w in out["worlds"].items():
        w["speedup_two_in_flight"] = one["max_ms_two_in_flight"] / w["max_ms_two_in_flight"]
        w["speedup_one_at_a_time"] = one["max_ms_one_at_a_time"] / w["max_ms_one_at_a_time"]
    json.dump(out, sys.stdout, indent=1)
    print()
    for world, w in out["worlds"].items():
        print(f"world {world}: slowest share {w['max_ms_two_in_flight']:.3f} ms in flight ({w['speedup_two_in_flight']:.2f}x), "
              f"{w['max_ms_one_at_a_time']:.3f} ms alone ({w['speedup_one_at_a_time']:.2f}x), work max/mean "
              f"{w['work_max_over_mean']:.4f}", file=sys.stderr)


if __name__ == "__main__":
    main()
