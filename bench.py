#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + secondary) of the render hot path on
fixtures/ASCII/Test5.txt at 1920x1080, 16 spp, max-depth 5 (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --config cfg3|cfg4|cfg5        (the other BASELINE.json configurations)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is one frame.  With N > 1 each rank renders the interleaved 16x16 tiles it owns from its own
replica of the scene and the frame is gathered to rank 0 (RCCL); the gather is inside the timed region.
Two frames are kept in flight (mirt_render_tiles_begin / _end on two streams): frame i + 1 is enqueued
before the host waits for frame i, so the launch latencies of one frame's small deep-level launches
are covered by the other frame's work.  Every frame of the timed region is complete, gather included,
before the clock stops.  --frames-in-flight 1 renders them one after the other.
Rank 0 prints one JSON line.  `value` counts the rays all ranks traced per second of the slowest
rank's wall time, with the scene already resident in HBM.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per ray, fixed record sizes: 2 ray records + 2 hit records + the BVH nodes and
# primitive records the ray tests (N_node, N_prim are measured by the engine's counters)
RAY_B, HIT_B, NODE_B, PRIM_B = 64, 32, 64, 128
HBM_PEAK_GBS = 8000.0


def bytes_per_ray(n_node, n_prim):
    return 2 * RAY_B + 2 * HIT_B + n_node * NODE_B + n_prim * PRIM_B


def cpu_baseline(scene_name, width, height, spp, depth, rays_per_sample):
    """Times the CPU path on a bounded sample of the same workload (rank 0, N = 1 only).
    Prefers the unmodified reference CLI built by oracle/Makefile (kind "reference"); without it the
    CPU restatement in oracle/ is timed (kind "port")."""
    cores = int(os.environ.get("OMP_NUM_THREADS") or min(os.cpu_count() or 1, 16))
    ref = os.path.join(ROOT, "oracle", "_ref", "raytracer_ref")
    sample = f"{scene_name} {width}x{height} {spp} spp max-depth {depth} (the bench frame at 1/4 of its pixels)"
    if os.path.exists(ref):
        with tempfile.TemporaryDirectory() as tmp:
            env = dict(os.environ, OMP_NUM_THREADS=str(cores))
            cmd = [ref, "--scene", os.path.join("ASCII", f"{scene_name}.txt"), "--resolution", str(width), str(height),
                   "--samples", str(spp), "--max-depth", str(depth), "--output", os.path.join(tmp, "cpu.ppm")]
            r = subprocess.run(cmd, cwd=os.path.join(ROOT, "fixtures"), env=env, capture_output=True, text=True)
            m = re.search(r"seconds=([0-9.]+)\s+Render time|Render time.*?seconds=([0-9.]+)", r.stdout)
            if r.returncode == 0 and m:
                sec = float(m.group(1) or m.group(2))  # the reference's own "Render time" line
                if sec > 0:
                    return {"value": width * height * spp * rays_per_sample / sec / 1e6, "unit": "Mrays/s",
                            "cores": cores, "kind": "reference", "seconds": sec,
                            "sample": sample + "; ray count = samples x the engine's measured rays per sample"}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        from raytracer_amd import HostScene, default_config
        hs = HostScene(os.path.join(ROOT, "fixtures", "ASCII", f"{scene_name}.txt"))
        cam, _ = hs.camera(0)
        cfg = default_config(max_ray_depth=depth)
        t0 = time.perf_counter()
        _, st = O.render(hs.desc, cam, cfg, width, height, spp, O.RNG_KEYED)
        sec = time.perf_counter() - t0
        return {"value": (st["primary_rays"] + st["secondary_rays"]) / sec / 1e6, "unit": "Mrays/s",
                "cores": O.lib().oracle_max_threads(), "kind": "port", "seconds": sec, "sample": sample}
    except Exception as e:  # the baseline is informative; never fail the bench on it
        return {"value": None, "unit": "Mrays/s", "cores": cores, "kind": "unavailable", "sample": str(e)}


# BASELINE.json configs[1..4]: argument defaults per preset (explicit flags still override)
PRESETS = {
    "cfg2": dict(scene="Test5", width=1920, height=1080, spp=16, max_depth=5),
    "cfg3": dict(scene="Test5", width=1920, height=1080, spp=16, max_depth=8, soft_shadows=16, glossy_reflection=8),
    "cfg4": dict(scene="synthetic:1000000", width=3840, height=2160, spp=16, max_depth=5),
    "cfg5": dict(scene="Test7_glass", width=1920, height=1080, spp=64, max_depth=8, soft_shadows=16,
                 glossy_reflection=4, depth_of_field=[0.95, 16.12]),
}


def hbm_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` from the committed PMC summary (profiles/traffic.json, written by
    profiles/traffic_summary.py from rocprofv3 FETCH_SIZE / WRITE_SIZE passes), if it is for this workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        if t.get("workload") == workload and kernel in t.get("kernels", {}):
            return t["kernels"][kernel]["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=sorted(PRESETS), default="cfg2")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default=None, help="scene under fixtures/ASCII, or synthetic:<primitives>")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--max-depth", type=int, default=None)
    ap.add_argument("--soft-shadows", type=int, default=None)
    ap.add_argument("--glossy-reflection", type=int, default=None)
    ap.add_argument("--depth-of-field", type=float, nargs=2, default=None, metavar=("FSTOP", "FOCUS"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, choices=(1, 2), default=2)
    args = ap.parse_args()
    for k, v in PRESETS[args.config].items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    if args.glossy_reflection is None:
        args.glossy_reflection = 0

    import torch
    import torch.distributed as dist

    from raytracer_amd import DeviceScene, HostScene, default_config
    from raytracer_amd import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the N > 1 control flow on a one-GPU box: all ranks render on GPU 0 and the gather
    # goes through host memory with gloo (RCCL refuses two ranks on one device).  Not a measurement.
    rehearsal = bool(os.environ.get("MIRT_BENCH_ONE_GPU"))
    if rehearsal:
        local_rank = 0
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the engine has no CPU path")
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    if args.scene.startswith("synthetic:"):
        hs = HostScene.synthetic(int(args.scene.split(":")[1]))
    else:
        hs = HostScene(os.path.join(ROOT, "fixtures", "ASCII", f"{args.scene}.txt"))
    ds = DeviceScene(hs, local_rank)
    cam, _ = hs.camera(0, dof=tuple(args.depth_of_field) if args.depth_of_field else None)
    over = dict(max_ray_depth=args.max_depth, glossy_samples=args.glossy_reflection)
    if args.soft_shadows is not None:
        over["shadow_samples"] = args.soft_shadows
    # per-kernel HIP events (for the roofline entry) cost ~10 us per launch and, with two frames in
    # flight, would time kernels that share the GPU with the other frame's: the timed region runs
    # without them, and the kernels are timed on frames rendered one at a time right after it
    over["kernel_timing"] = 0
    cfg = default_config(**over)
    W, H, spp, tile = args.width, args.height, args.spp, D.TILE

    depth = args.frames_in_flight
    if world > 1:
        bufs = [D.local_buffer(W, H, tile, tile, world, device) for _ in range(depth)]
    else:
        bufs = [torch.empty(W * H * 3, dtype=torch.float64, device=device) for _ in range(depth)]
    streams = [torch.cuda.Stream(device) for _ in range(depth)]
    frames = [None] * depth
    gather_to = [0]  # rank that receives the frame; None: all-gather (fallback if the backend lacks gather)

    def gather(src):
        try:
            return D.gather_frame(src, W, H, tile, tile, dst=gather_to[0])
        except (RuntimeError, NotImplementedError) as e:
            if gather_to[0] is None:
                raise
            if rank == 0:
                print(f"bench: gather to rank 0 failed ({e}); using all-gather", file=sys.stderr)
            gather_to[0] = None  # every rank fails the same way on an unsupported collective
            return D.gather_frame(src, W, H, tile, tile, dst=None)

    def begin(slot):
        """Enqueues one frame (this rank's tiles, then the gather) on the slot's stream."""
        s = streams[slot]
        if world > 1:
            ds.render_tiles_begin(cam, cfg, W, H, spp, tile, tile, rank, world, bufs[slot].data_ptr(), s.cuda_stream, slot)
            with torch.cuda.stream(s):
                frames[slot] = gather(bufs[slot].cpu() if rehearsal else bufs[slot])
        else:
            ds.render_tiles_begin(cam, cfg, W, H, spp, W, H, 0, 1, bufs[slot].data_ptr(), s.cuda_stream, slot)
            frames[slot] = bufs[slot].view(H, W, 3)

    def end(slot):
        """Waits for the slot's frame; returns its stats."""
        st = ds.render_tiles_end(slot)
        if st["frame_attempts"] > 1 and world > 1:
            # the frame was redone after its gather had been enqueued, on this rank only: the sizing
            # frame below exists so that this cannot happen once the collectives have started
            raise RuntimeError("a ray queue overflowed during the run; raise MIRT_RAY_QUEUE / MIRT_SHADOW_QUEUE")
        return st

    def run(n, depth=depth):
        """n frames, `depth` of them in flight; returns the summed stats."""
        acc = None
        for i in range(n + depth):
            slot = i % depth
            if i >= depth:
                st = end(slot)
                acc = st if acc is None else {k: acc[k] + st[k] for k in st}
            if i < n:
                begin(slot)
        for s in streams:
            s.synchronize()
        return acc

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    # sizing frame, no collective: allocates the working memory and, should a queue overflow, finds
    # the finer launch split this frame shape needs (the scene handle remembers it) on every rank by itself
    for slot in range(depth):
        if world > 1:
            ds.render_tiles_begin(cam, cfg, W, H, spp, tile, tile, rank, world, bufs[slot].data_ptr(),
                                  streams[slot].cuda_stream, slot)
        else:
            ds.render_tiles_begin(cam, cfg, W, H, spp, W, H, 0, 1, bufs[slot].data_ptr(), streams[slot].cuda_stream, slot)
        ds.render_tiles_end(slot)
    if args.warmup:
        run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    acc = run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    cfg.kernel_timing = 1
    n_timed = args.steps if world == 1 else 1
    st = run(n_timed, depth=1)
    for k in ("trace_kernel_ms", "shadow_kernel_ms", "trace_kernel_launches", "shadow_kernel_launches"):
        acc[k] = st[k] * (args.steps / n_timed)
    barrier()

    keys = ["primary_rays", "secondary_rays", "shadow_segments", "node_tests", "prim_tests", "shadow_node_tests",
            "shadow_prim_tests", "trace_kernel_launches", "shadow_kernel_launches"]
    red_dev = "cpu" if rehearsal else device
    tot = torch.tensor([float(acc[k]) for k in keys] + [acc["trace_kernel_ms"], acc["shadow_kernel_ms"]],
                       dtype=torch.float64, device=red_dev)
    tmax = torch.tensor([elapsed, acc["trace_kernel_ms"], acc["shadow_kernel_ms"]], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tot = dict(zip(keys + ["trace_kernel_ms", "shadow_kernel_ms"], tot.tolist()))
    elapsed = float(tmax[0])

    if rank == 0:
        steps = args.steps
        rays_ps = tot["primary_rays"] + tot["secondary_rays"]
        rays_all = rays_ps + tot["shadow_segments"]
        # dominant kernel = the kind with the larger summed HIP-event time on this run
        kinds = {
            "k_trace": (rays_ps, tot["node_tests"], tot["prim_tests"], tot["trace_kernel_ms"], tot["trace_kernel_launches"]),
            "k_shadow": (tot["shadow_segments"], tot["shadow_node_tests"], tot["shadow_prim_tests"],
                         tot["shadow_kernel_ms"], tot["shadow_kernel_launches"]),
        }
        name = max(kinds, key=lambda k: kinds[k][3])
        q, nn, npr, ms, launches = kinds[name]
        b_ray = bytes_per_ray(nn / max(q, 1), npr / max(q, 1))
        avg_ms = ms / max(launches, 1)
        achieved = (q / max(launches, 1)) * b_ray / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        workload = (f"{args.scene} {W}x{H} {spp}spp max-depth {args.max_depth} motion-blur on"
                    + (f" glossy {args.glossy_reflection}" if args.glossy_reflection else "")
                    + (f" soft-shadows {args.soft_shadows}" if args.soft_shadows is not None else "")
                    + (f" dof {args.depth_of_field[0]} {args.depth_of_field[1]}" if args.depth_of_field else ""))
        traffic = hbm_traffic(name, workload) if world == 1 else None
        line = {
            "metric": f"Mrays/s (primary+secondary) at {W}x{H}x{spp}spp depth {args.max_depth}",
            "value": rays_ps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" if not (world > 1 and rehearsal) else "synthetic (REHEARSAL: all ranks on one GPU, not a measurement)",
            "config": {"workload": workload, "preset": args.config, "tile": tile if world > 1 else None,
                       "parallelism": f"tiles/{world}", "frames_in_flight": depth},
            "mrays_all_per_s": rays_all / elapsed / 1e6,
            "rays_per_sample": rays_ps / max(tot["primary_rays"], 1),
            "shadow_segments_per_sample": tot["shadow_segments"] / max(tot["primary_rays"], 1),
            "roofline": {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "bytes_per_ray": b_ray, "node_tests_per_ray": nn / max(q, 1),
                         "prim_tests_per_ray": npr / max(q, 1), "rays_per_launch": q / max(launches, 1),
                         "avg_launch_ms": avg_ms,
                         "timed_on": f"{n_timed} frame(s) rendered one at a time after the timed region",
                         "note": "algorithmic bytes charge every node and primitive test as memory traffic; a scene "
                                 "that stays in the scalar cache / L2 can therefore exceed the HBM peak -- traffic "
                                 "is the measured HBM bytes per launch"},
        }
        if world == 1 and not args.no_cpu_baseline and args.config == "cfg2":
            line["cpu_baseline"] = cpu_baseline(args.scene, W // 2, H // 2, spp, args.max_depth, line["rays_per_sample"])
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
