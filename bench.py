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
# Kernel arguments in device memory instead of host memory (read by the HIP runtime when it starts, hence before
# torch): the kernels read their argument block through the scalar cache wherever they need it, and a frame is a
# dozen launches, some of them tens of microseconds long.  Measured: 8-way shares 0.657 -> 0.636 ms, whole frames +0.5 %.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

# algorithmic bytes per ray, fixed record sizes: 2 ray records + 2 hit records + the BVH nodes and
# primitive records the ray tests (N_node, N_prim are measured by the engine's counters)
RAY_B, HIT_B, NODE_B, PRIM_B = 64, 32, 64, 128
HBM_PEAK_GBS = 8000.0
# vector issue capacity of the chip: 256 CUs x 4 SIMDs at 2.4 GHz.  A 64-lane vector instruction
# occupies its SIMD for 2 cycles, a double-precision one for 4 (157 / 79 TFLOP/s vector peaks).
SIMDS, CLOCK_GHZ = 1024, 2.4
CYC_F64, CYC_OTHER = 4.0, 2.0


def bytes_per_ray(n_node, n_prim):
    return 2 * RAY_B + 2 * HIT_B + n_node * NODE_B + n_prim * PRIM_B


def cpu_baseline(args, rays_per_sample, shadow_per_sample):
    """Times the CPU path on the same workload (rank 0, N = 1 only): the unmodified reference program
    built by oracle/Makefile (kind "reference") renders the whole bench frame where that takes seconds
    (Test5 at 16 spp), the frame at a stated fraction of its pixels where it would take minutes; the
    generated cloud, which the reference could only read as a gigabyte of text, and a box without the
    reference build time the CPU restatement in oracle/ instead (kind "port")."""
    cores = int(os.environ.get("OMP_NUM_THREADS") or min(os.cpu_count() or 1, 16))
    ref = os.path.join(ROOT, "oracle", "_ref", "raytracer_ref")
    scene_name, spp, depth = args.scene, args.spp, args.max_depth
    # bounded sample: the frame is shrunk per axis until its rays + shadow segments (as the engine counted
    # them) come to about 20 s of the reference at the rate it shows on Test5 (35 M segments/s on 16 cores)
    work = args.width * args.height * spp * (rays_per_sample + shadow_per_sample)
    shrink = 1
    while work / (shrink * shrink) > 7e8 * cores / 16 and shrink < 16:
        shrink *= 2
    print(f"bench.py: timing the CPU baseline (1/{shrink * shrink} of the frame's pixels) ...", file=sys.stderr, flush=True)
    width, height = args.width // shrink, args.height // shrink
    sample = (f"{scene_name} {width}x{height} {spp} spp max-depth {depth}: "
              + ("the whole bench frame" if shrink == 1 else f"the bench frame at 1/{shrink * shrink} of its pixels"))
    if os.path.exists(ref) and not scene_name.startswith("synthetic:"):
        with tempfile.TemporaryDirectory() as tmp:
            env = dict(os.environ, OMP_NUM_THREADS=str(cores))
            cmd = [ref, "--scene", os.path.join("ASCII", f"{scene_name}.txt"), "--resolution", str(width), str(height),
                   "--samples", str(spp), "--max-depth", str(depth), "--output", os.path.join(tmp, "cpu.ppm")]
            if args.glossy_reflection:
                cmd += ["--glossy-reflection", str(args.glossy_reflection)]
            if args.soft_shadows is not None:
                cmd += ["--soft-shadows", str(args.soft_shadows)]
            if args.depth_of_field:
                cmd += ["--depth-of-field", str(args.depth_of_field[0]), str(args.depth_of_field[1])]
            r = subprocess.run(cmd, cwd=os.path.join(ROOT, "fixtures"), env=env, capture_output=True, text=True)
            m = re.search(r"seconds=([0-9.]+)\s+Render time|Render time.*?seconds=([0-9.]+)", r.stdout)
            if r.returncode == 0 and m:
                sec = float(m.group(1) or m.group(2))  # the reference's own "Render time" line
                if sec > 0:
                    return {"value": width * height * spp * rays_per_sample / sec / 1e6, "unit": "Mrays/s",
                            "cores": cores, "kind": "reference", "seconds": sec,
                            "sample": sample + "; timed by the reference's own 'Render time' line; rays = samples x "
                                      "the rays per sample the engine counted on the same workload (the reference "
                                      "counts none)"}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        from raytracer_amd import HostScene, default_config
        if scene_name.startswith("synthetic:"):
            hs = HostScene.synthetic(int(scene_name.split(":")[1]))
            win = (args.width // 2 - 240, args.height // 2 - 135, 480, 270)  # a window of the frame
            sample = f"{scene_name} {args.width}x{args.height} {spp} spp max-depth {depth}: window {win} of the bench frame"
        else:
            hs = HostScene(os.path.join(ROOT, "fixtures", "ASCII", f"{scene_name}.txt"))
            win = (0, 0, width, height)
        cam, _ = hs.camera(0, dof=tuple(args.depth_of_field) if args.depth_of_field else None)
        over = dict(max_ray_depth=depth, glossy_samples=args.glossy_reflection or 0)
        if args.soft_shadows is not None:
            over["shadow_samples"] = args.soft_shadows
        cfg = default_config(**over)
        full = (args.width, args.height) if scene_name.startswith("synthetic:") else (width, height)
        t0 = time.perf_counter()
        _, st = O.render_window(hs.desc, cam, cfg, full[0], full[1], spp, *win, O.RNG_KEYED)
        sec = time.perf_counter() - t0
        return {"value": (st["primary_rays"] + st["secondary_rays"]) / sec / 1e6, "unit": "Mrays/s",
                "cores": O.lib().oracle_max_threads(), "kind": "port", "seconds": sec, "sample": sample}
    except Exception as e:  # the baseline is informative; never fail the bench on it
        return {"value": None, "unit": "Mrays/s", "cores": cores, "kind": "unavailable", "sample": str(e)}


def caller_times(args, hs, ds, cam, cfg, scene_create, load_scene, device, rays_per_frame):
    """What a caller of the drop-in sees around the frames `value` is made of (none of it is in `value`):
    scene set-up with either BVH builder (flatten on the host, then mirt_scene_create), one frame through
    mirt_render into pageable host memory, and the wall clock of the whole command line."""
    from raytracer_amd import DeviceScene
    out = {"scene_create_ms": scene_create}
    try:
        other = "host" if args.builder == "device" else "device"
        h2, flatten_ms = load_scene(other == "host")
        d2 = DeviceScene(h2, device, builder=other)
        i2 = d2.info()
        scene_create[other] = {"flatten_ms": flatten_ms, "create_ms": i2["create_ms"], "tree_ms": i2["tree_ms"]}
        d2.close()
        h2.close()
    except Exception as e:  # the device builder refuses scenes with planes
        scene_create["device" if args.builder == "host" else "host"] = {"unavailable": str(e)[:200]}
    cfg.kernel_timing = 0
    ds.render(cam, cfg, args.width, args.height, args.spp)  # sizes the staging memory
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        ds.render(cam, cfg, args.width, args.height, args.spp)
    out["host_frame_ms"] = (time.perf_counter() - t0) / n * 1e3
    out["host_frame_mrays_per_s"] = rays_per_frame / out["host_frame_ms"] / 1e3
    cli = os.path.join(ROOT, "raytracer_amd", "bin", "raytracer")
    if os.path.exists(cli) and not args.scene.startswith("synthetic:"):
        with tempfile.TemporaryDirectory() as tmp:
            cmd = [cli, "--scene", os.path.join("ASCII", f"{args.scene}.txt"), "--resolution", str(args.width),
                   str(args.height), "--samples", str(args.spp), "--max-depth", str(args.max_depth), "--binary-ppm",
                   "--output", os.path.join(tmp, "cli.ppm")]
            if args.glossy_reflection:
                cmd += ["--glossy-reflection", str(args.glossy_reflection)]
            if args.soft_shadows is not None:
                cmd += ["--soft-shadows", str(args.soft_shadows)]
            if args.depth_of_field:
                cmd += ["--depth-of-field", str(args.depth_of_field[0]), str(args.depth_of_field[1])]
            t0 = time.perf_counter()
            r = subprocess.run(cmd, cwd=os.path.join(ROOT, "fixtures"), capture_output=True, text=True)
            out["cli_wall_s"] = time.perf_counter() - t0 if r.returncode == 0 else None
            m = re.search(r"Render time seconds=([0-9.]+)", r.stdout + r.stderr)
            out["cli_render_s"] = float(m.group(1)) if m else None
    return out


# BASELINE.json configs[1..4]: argument defaults per preset (explicit flags still override)
PRESETS = {
    "cfg2": dict(scene="Test5", width=1920, height=1080, spp=16, max_depth=5),
    "cfg3": dict(scene="Test5", width=1920, height=1080, spp=16, max_depth=8, soft_shadows=16, glossy_reflection=8),
    "cfg4": dict(scene="synthetic:1000000", width=3840, height=2160, spp=16, max_depth=5),
    "cfg5": dict(scene="Test7_glass", width=1920, height=1080, spp=64, max_depth=8, soft_shadows=16,
                 glossy_reflection=4, depth_of_field=[0.95, 16.12]),
}


def counter_model(kernel, workload):
    """What the committed counter passes say about `kernel` on this workload (profiles/kernel_model.json,
    written by profiles/kernel_model.py from rocprofv3 --pmc runs of this same command): HBM bytes per
    launch (FETCH_SIZE / WRITE_SIZE), vector instructions per ray or shadow segment and the share of them
    that is double precision.  None when no pass was collected for the workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "kernel_model.json")) as f:
            t = json.load(f)
        return t[workload]["kernels"][kernel]
    except (OSError, ValueError, KeyError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=sorted(PRESETS), default="cfg2")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default=None, help="scene under fixtures/ASCII, or synthetic:<primitives>")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--max-depth", type=int, default=None)
    ap.add_argument("--soft-shadows", type=int, default=None)
    ap.add_argument("--glossy-reflection", type=int, default=None)
    ap.add_argument("--depth-of-field", type=float, nargs=2, default=None, metavar=("FSTOP", "FOCUS"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, choices=(1, 2, 3, 4), default=None,
                    help="default: 2; 1 for the generated cloud, whose frame is long enough that a second one in flight "
                         "only competes for the caches")
    ap.add_argument("--builder", choices=("host", "device"), default=None,
                    help="where the BVH is built (default: host; the generated cloud: device)")
    ap.add_argument("--no-caller-times", action="store_true",
                    help="skip the figures outside `value`: the other builder's scene creation, mirt_render into host memory, the CLI")
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

    synthetic = args.scene.startswith("synthetic:")
    if args.frames_in_flight is None:
        args.frames_in_flight = 1 if synthetic else 2
    if args.builder is None:
        # the generated cloud has no planes: its tree is built on the GPU (12 ms against more than a second
        # on the host); the bundled scenes keep the reference's tree
        args.builder = "device" if synthetic else "host"

    def load_scene(build_tree):
        t0 = time.perf_counter()
        if synthetic:
            h = HostScene.synthetic(int(args.scene.split(":")[1]), build_tree=build_tree)
            flatten_ms = h.flatten_ms  # (the generation of the cloud itself is not part of a scene's set-up)
        else:
            h = HostScene(os.path.join(ROOT, "fixtures", "ASCII", f"{args.scene}.txt"), build_tree=build_tree)
            flatten_ms = (time.perf_counter() - t0) * 1e3  # parse + flatten
        return h, flatten_ms

    free_before = torch.cuda.mem_get_info(device)[0]
    hs, flatten_ms = load_scene(args.builder == "host")
    # the process's first creation also loads the code object and sets up the runtime's memory pools:
    # one creation goes unrecorded, so that create_ms is what every later scene of the process costs
    DeviceScene(hs, local_rank, builder=args.builder).close()
    ds = DeviceScene(hs, local_rank, builder=args.builder)
    scene_info = ds.info()
    scene_create = {args.builder: {"flatten_ms": flatten_ms, "create_ms": scene_info["create_ms"],
                                   "tree_ms": scene_info["tree_ms"]}}
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
    for st_ in streams:  # the buffers were made (and zero-filled) on the current stream
        st_.wait_stream(torch.cuda.current_stream(device))
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
    device_memory_mb = (free_before - torch.cuda.mem_get_info(device)[0]) / 2**20  # scene, queues, frames in flight
    t0 = time.perf_counter()
    acc = run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    # latency of one frame: the same frames one at a time (enqueue, wait, next)
    n_timed = args.steps if world == 1 else 1
    t0 = time.perf_counter()
    run(n_timed, depth=1)
    barrier()
    latency = (time.perf_counter() - t0) / n_timed
    cfg.kernel_timing = 1
    st = run(n_timed, depth=1)
    for k in ("trace_kernel_ms", "shadow_kernel_ms", "trace_kernel_launches", "shadow_kernel_launches"):
        acc[k] = st[k] * (args.steps / n_timed)
    barrier()

    keys = ["primary_rays", "secondary_rays", "shadow_segments", "node_tests", "prim_tests", "shadow_node_tests",
            "shadow_prim_tests", "trace_kernel_launches", "shadow_kernel_launches"]
    red_dev = "cpu" if rehearsal else device
    tot = torch.tensor([float(acc[k]) for k in keys] + [acc["trace_kernel_ms"], acc["shadow_kernel_ms"]],
                       dtype=torch.float64, device=red_dev)
    tmax = torch.tensor([elapsed, acc["trace_kernel_ms"], acc["shadow_kernel_ms"], latency], dtype=torch.float64,
                        device=red_dev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tot = dict(zip(keys + ["trace_kernel_ms", "shadow_kernel_ms"], tot.tolist()))
    elapsed, latency = float(tmax[0]), float(tmax[3])

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
        # The bound that binds is vector issue, most of it double precision: the scenes are a few KB
        # and sit in the scalar cache, so charging every node and primitive test as memory traffic
        # (the algorithmic-bytes figure) overshoots what reaches HBM several times.  `frac` prices the
        # kernel's vector instructions -- counted per ray / shadow segment by the committed counter
        # passes, times the rays this run counted -- against the chip's issue capacity over the
        # measured launch time; hbm_frac does the same with the measured HBM bytes.
        model = counter_model(name, workload) if world == 1 else None
        units_per_launch = q / max(launches, 1)
        traffic = issue = hbm_frac = None
        if model and avg_ms > 0:
            traffic = model["hbm_bytes_per_unit"] * units_per_launch
            hbm_frac = traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            cyc = model["valu_per_unit"] * (model["fp64_share"] * CYC_F64 + (1.0 - model["fp64_share"]) * CYC_OTHER)
            issue = cyc * units_per_launch / (avg_ms * 1e-3) / 1e9  # G SIMD-cycles of vector issue per second
        peak_issue = SIMDS * CLOCK_GHZ
        line = {
            "metric": f"Mrays/s (primary+secondary) at {W}x{H}x{spp}spp depth {args.max_depth}",
            "value": rays_ps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "frame_latency_ms": latency * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": ("synthetic" if args.scene.startswith("synthetic:") else "fixture")
                    + (" (REHEARSAL: all ranks on one GPU, not a measurement)" if world > 1 and rehearsal else ""),
            "config": {"workload": workload, "preset": args.config, "tile": tile if world > 1 else None,
                       "parallelism": f"tiles/{world}", "frames_in_flight": depth, "bvh_builder": args.builder,
                       "traversal": {1: "packet", 2: "lane", 3: "ordered", 4: "wide"}[scene_info["traversal"]],
                       "shadow_any_hit": bool(scene_info["shadow_any_hit"])},
            "mrays_all_per_s": rays_all / elapsed / 1e6,
            "rays_per_sample": rays_ps / max(tot["primary_rays"], 1),
            "shadow_segments_per_sample": tot["shadow_segments"] / max(tot["primary_rays"], 1),
            "device_memory_mb": device_memory_mb,
            "roofline": {"bound": "valu_fp64", "kernel": name, "achieved": issue, "peak": peak_issue,
                         "unit": "G SIMD-cycles/s of vector issue",
                         "frac": issue / peak_issue if issue is not None else None,
                         "traffic": traffic, "hbm_frac": hbm_frac, "hbm_peak_gbs": HBM_PEAK_GBS,
                         "algorithmic_gbs": achieved, "bytes_per_ray": b_ray,
                         "node_tests_per_ray": nn / max(q, 1), "prim_tests_per_ray": npr / max(q, 1),
                         "rays_per_launch": units_per_launch, "avg_launch_ms": avg_ms,
                         "valu_per_ray": model["valu_per_unit"] if model else None,
                         "fp64_share": model["fp64_share"] if model else None,
                         "counters_from": model["source"] if model else None,
                         "timed_on": f"{n_timed} frame(s) rendered one at a time after the timed region",
                         "note": f"ms_per_step is the interval between frames with {depth} in flight, frame_latency_ms one "
                                 "frame alone; frac = vector issue cycles (double precision 4, other 2 per wave "
                                 "instruction) / (1024 SIMDs x 2.4 GHz x launch time); hbm_frac = measured HBM bytes "
                                 "/ launch time / 8 TB/s; algorithmic_gbs charges every node and primitive test as "
                                 "memory traffic and is not a fraction of anything (the scene stays in cache)"},
        }
        if world == 1 and not args.no_caller_times:
            line["caller"] = caller_times(args, hs, ds, cam, cfg, scene_create, load_scene, local_rank, rays_ps / steps)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, line["rays_per_sample"], line["shadow_segments_per_sample"])
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
