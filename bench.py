#!/usr/bin/env python3
"""Benchmark of the render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one full render of the headline frame (book-1 final scene, 1200x800, spp 500,
depth 50, 484 spheres, seed 0xE7): clear + trace + resolve on the GPU with scene and camera
already resident; nothing crosses PCIe inside the timed region except the 16-byte counter
read-back.  Steps are kept two deep in flight (two contexts, each with its own stream and
buffers, used in turn): the tail of a frame overlaps the start of the next one.  All K steps
start and complete inside the timed region; `latency_ms_per_step` is one step on its own.
With N > 1 - one process per GPU: either started by torch.distributed.run, or by this script
itself when it is called plainly as `python bench.py --gpus N` (it then starts the N workers
before touching any GPU and relays rank 0's line) - the same frame is sharded by interleaved
row tiles and each step ends with the RCCL gather of the 8-bit rows, so total work is fixed
("strong" scaling).

Prints ONE JSON line on rank 0.  value = ray segments (world.hit queries) of all ranks / wall
seconds / 1e6.  roofline: algorithmic FLOP (18 per ray-sphere test, 18*N+65 per segment) over
the trace kernel's own duration, measured with HIP events on the stream it runs on in three
un-overlapped steps after the timed region (the median of the three).  The same three steps
are then repeated with the camera rays' first tests switched off (frac_without_first_test),
with the interval test alone, with the slab test in their place, and with the optional bundle
test; none of those enters `value` or `frac`.
cpu_baseline: the FP64 CPU restatement of the reference (oracle/, "port") timed on this box's
host cores on a bounded sample of the same frame.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X vector FP32 peak: 256 CU x 2.4 GHz x 256 FLOP/clk/CU
FLOP_PER_TEST = 18             # hitrecord.nim:56-61: oc 3, half_b 5, |oc|^2 5, r*r 1, c 1, disc 3
FLOP_PER_SEGMENT_EXTRA = 65    # a = |dir|^2 once, hit epilogue, scatter, sky (amortised)
FLOP_PER_TEST_EXECUTED = 14    # what the kernel's screen really issues per sphere: 7 packed FMAs per pair
# camera rays (one pixel's samples per wave) see a cheaper necessary condition per (ray, sphere) pair first, and only the
# groups of four spheres that pass it in some lane go on to the screen (screen_pass_fraction() below):
# one subtraction; two FMAs; nothing; one AND per group of four spheres (plus two subtractions for the pairs of the groups
# whose word has the ray's bit: mask_word_pass in the same file)
FLOP_PER_CAMERA_TEST = {"interval": 1, "slab": 4, "none": 0, "mask": 0.25}
SCREEN_NAMES = {0: "none", 1: "slab", 2: "interval", 3: "bundle", 4: "mask"}   # rtow_stats.camera_screen

WORKLOADS = {
    # name: (width, height, spp, depth, half_extent, lambert_below, metal_below)
    "C2": (1200, 800, 500, 50, 11, 0.8, 0.95),
    "C3": (1200, 800, 500, 50, 11, 0.0, 0.5),     # every small sphere metal or glass
    "C5": (3840, 2160, 1024, 50, 50, 0.8, 0.95),  # ~10k spheres, brute force
}


def screen_pass_fraction(workload, spp, screen):
    """Share of a PRIMARY scan's groups that go on from the camera rays' first test to the 7-FMA screen, measured per
    workload and per kind of test with the instrumented build (profiles/r*_screen_pass_fraction.json).  None for a
    configuration it was not measured on (another sample count changes which rays share a wavefront)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_screen_pass_fraction.json")))
    if not found or spp != WORKLOADS[workload][2]:
        return None
    table = json.load(open(found[-1]))
    if screen == "mask_word_pass":
        return table.get(screen, {}).get(workload, 0.0)
    return table.get(screen, {}).get(workload)


def hbm_traffic(workload, spp, world):
    """HBM bytes per launch of the trace kernel from the committed rocprofv3 PMC passes
    (FETCH_SIZE doubled as gfx950 requires, WRITE_SIZE as is); only valid for the configuration it
    was measured on."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))   # the latest round's passes
    if world != 1 or spp != WORKLOADS[workload][2] or not found:
        return None
    path = found[-1]
    t = json.load(open(path))
    if t.get("workload") != workload:
        return None
    return t["fetch_bytes"] + t["write_bytes"]


def cpu_baseline(width, height, depth, seed):
    """Time the FP64 restatement of the reference on a bounded sample: 16 rows spread over the
    frame.  Single thread first (the reference is single-threaded), then all host cores.  Also the
    reference's own CPU-sized case in full: 400x225, spp 10, depth 50 (the first of the benchmark configurations), one thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    arr, n, _, _ = O.random_scene()
    t0 = time.perf_counter()
    _, _, _, seg_c1 = O.render_ref64(arr, n, O.book_camera(400 / 225), 400, 225, 10, 50, seed=seed, threads=1)
    t_c1 = time.perf_counter() - t0
    cam = O.book_camera(width / height)
    rows = list(range(height // 32, height, height // 16))[:16]
    spp = 256
    t0 = time.perf_counter()
    _, _, _, seg1 = O.render_ref64(arr, n, cam, width, height, spp, depth, seed=seed, rows=rows, threads=1)
    t1 = time.perf_counter() - t0
    cores = max(1, min(16, os.cpu_count() or 1))
    t0 = time.perf_counter()
    _, _, _, segm = O.render_ref64(arr, n, cam, width, height, 2 * spp, depth, seed=seed, rows=rows, threads=cores)
    tm = time.perf_counter() - t0
    return {
        "value": seg1 / t1 / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
        "sample": f"oracle/ref64.c (FP64 restatement of the Nim reference, which cannot be built here), "
                  f"{len(rows)} rows spread over the {width}x{height} frame, spp {spp}, depth {depth}, 1 thread, "
                  f"{seg1} segments in {t1:.1f} s",
        "value_all_cores": segm / tm / 1e6, "cores_all": cores,
        "sample_all_cores": f"same rows, spp {2 * spp}, {cores} pthreads, {segm} segments in {tm:.1f} s",
        "c1_full": {"workload": "C1: 400x225, spp 10, depth 50, 484 spheres, whole frame", "cores": 1,
                    "seconds": t_c1, "segments": seg_c1, "value": seg_c1 / t_c1 / 1e6, "unit": "Mrays/s"},
    }


def launch_workers(n_gpus):
    """Run this script as n_gpus worker processes (one per GPU) under torch.distributed.run and relay the
    result: rank 0's JSON line goes to stdout, everything else the workers print to stderr.  Returns the
    exit status of the launcher.  The calling process must not have initialised the GPU."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs between processes on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    n_lines = 0
    for line in proc.stdout:
        text = line.strip()
        is_result = False
        if text.startswith("{"):
            try:
                is_result = "metric" in json.loads(text)
            except ValueError:
                is_result = False
        if is_result:
            n_lines += 1
            print(text, flush=True)
        elif text:
            print(text, file=sys.stderr, flush=True)
    rc = proc.wait()
    if rc == 0 and n_lines != 1:
        print(f"bench.py: the workers exited normally but printed {n_lines} result lines", file=sys.stderr)
        return 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (invalidates the headline number)")
    ap.add_argument("--pipeline", type=int, default=2, help="frames in flight (contexts used in turn); 1 = strictly serial steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks "
                         "(ranks share devices, rows are gathered through host memory); not a valid benchmark")
    ap.add_argument("--rehearse-collective", action="store_true",
                    help="with one rank: run the multi-GPU code path anyway (process group, RCCL all_gather of the rows "
                         "on the render stream, assembly, comparison with a whole-frame render); a check of the plumbing "
                         "on a one-GPU box, not a benchmark")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # Plain `python bench.py --gpus N`: this process becomes the launcher.  It starts one worker per GPU through
        # torch.distributed.run, passes rank 0's JSON line on and exits with the workers' status.  Nothing here has
        # touched a GPU or imported torch, and nothing is exec'ed: the workers are fresh child processes.
        raise SystemExit(launch_workers(args.gpus))

    if args.rehearse_collective and "RANK" not in os.environ:   # a one-rank job of its own
        import socket
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sock.getsockname()[1]))
        sock.close()
    # stdout carries ONE line, the result.  Libraries write there too (RCCL prints a version banner on rank 0 when its
    # first communicator is built), so for the length of the run file descriptor 1 points at stderr and the result
    # line goes to the saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        args.gpus = world
    sharded = world > 1 or args.rehearse_collective   # rows are dealt to ranks and gathered

    width, height, spp, depth, half_extent, lam, met = WORKLOADS[args.workload]
    if args.spp:
        spp = args.spp
    seed = 0xE7

    dist = torch = None
    device = local_rank
    if sharded:
        import torch
        import torch.distributed as dist
        if args.backend == "gloo":
            device = local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(device)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))

    from raytracinginoneweekend_amd import multigpu, raytracer as R

    world_scene = R.randomScene(halfExtent=half_extent, lambertBelow=lam, metalBelow=met)
    n_spheres = len(world_scene)
    tile = 4
    rows = multigpu.tile_rows(height, world, rank, tile) if sharded else None
    n_rows = height if rows is None else len(rows)

    # Frames in flight: PIPELINE contexts (each its own stream, accumulators and image) take the
    # steps in turn, so that the tail of one frame - a few wavefronts finishing their longest paths,
    # the resolve kernel, the gather of the rows - overlaps the start of the next frame.
    depth_in_flight = max(1, args.pipeline)
    renderers, gatherers, images, streams = [], [], [], []
    for _ in range(depth_in_flight):
        r = R.Renderer(device)
        r.setWorld(world_scene)
        r.setCamera(R.bookCamera(width / height))
        renderers.append(r)
        if sharded:
            g = multigpu.TileGatherer(height, width, world, rank, f"cuda:{device}" if args.backend == "nccl" else "cpu", tile,
                                      always_collective=args.rehearse_collective)
            gatherers.append(g)
            images.append(g.local if args.backend == "nccl"
                          else torch.empty((n_rows, width, 3), dtype=torch.uint8, device=f"cuda:{device}"))
            # the context's own stream as a torch stream: the gather of a frame is enqueued on it right behind the
            # frame's kernels, so no host round trip sits between a frame and its gather
            streams.append(torch.cuda.ExternalStream(r.stream(), device=f"cuda:{device}") if args.backend == "nccl" else None)

    def sync_all():
        if sharded:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    def enqueue(i):
        k = i % depth_in_flight
        renderers[k].renderAsync(width, height, spp, depth, seed, rows=rows,
                                 rgb_dev=images[k].data_ptr() if images else None)
        if sharded and args.backend == "nccl":
            with torch.cuda.stream(streams[k]):
                gatherers[k].gather()   # stream-ordered behind the frame: the collective and rank 0's row scatter

    def finish(i):
        k = i % depth_in_flight
        renderers[k].sync()   # the context's stream: the frame and (RCCL) the gather enqueued behind it
        st = renderers[k].stats()
        if sharded and args.backend != "nccl":
            gatherers[k].local[:n_rows].copy_(images[k])
            gatherers[k].gather()
        return st

    def run(n_steps):
        out = []
        for i in range(n_steps):
            if i >= depth_in_flight:
                out.append(finish(i - depth_in_flight))
            enqueue(i)
        for i in range(max(0, n_steps - depth_in_flight), n_steps):
            out.append(finish(i))
        return out

    if sharded:   # communicator set-up and the first collective stay out of the timed region even with --warmup 0
        for g in gatherers:
            g.gather()
        sync_all()
    run(args.warmup)
    sync_all()
    t0 = time.perf_counter()
    stats = run(args.steps)
    sync_all()
    elapsed = time.perf_counter() - t0
    segments = sum(st["ray_segments"] for st in stats)
    if sharded:   # the slowest rank's time, before anything is decided from it: all ranks must take the same branches below
        red_dev = f"cuda:{device}" if args.backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        s = torch.tensor([segments], dtype=torch.int64, device=red_dev)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        segments = int(s.item())

    # one frame at a time, for the kernel's own duration (launches that overlap stretch each other)
    # and for the latency of a single step
    def note(msg):   # long configurations: a line per phase, so that a watcher sees the run is alive
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    note(f"timed region done: {elapsed / max(1, args.steps) * 1e3:.1f} ms per step")
    n_serial = 3 if elapsed / max(1, args.steps) < 2.0 else 1     # frames of many seconds: one is enough
    mid = n_serial // 2
    serial, serial_wall = [], []
    for i in range(n_serial):
        t1 = time.perf_counter()
        enqueue(i)
        serial.append(finish(i))
        serial_wall.append((time.perf_counter() - t1) * 1e3)
        note(f"serial step {i + 1} of {n_serial}")
    serial_ms = sorted(serial_wall)[mid]                      # medians of three: one slow launch (clock
    trace_ms = [sorted(st["trace_ms"] for st in serial)[mid]]  # transients between phases) does not decide
    total_ms = [sorted(st["total_ms"] for st in serial)[mid]]
    clock_mhz = sorted(st.get("shader_clock_mhz", 0.0) for st in serial)[mid]   # what the chip sustained under the kernel
    # the same three steps with nothing in front of the camera rays' screen (RTOW_NO_SLAB_TEST is read at every launch
    # and selects the kernel instance without a first test): every ray-sphere pair then goes through the 7-FMA screen
    screen = SCREEN_NAMES.get(serial[-1].get("camera_screen", 0), "none")
    slab_used = screen != "none"
    no_slab_ms = None
    if slab_used:
        os.environ["RTOW_NO_SLAB_TEST"] = "1"
        plain = []
        for i in range(n_serial):
            enqueue(i)
            plain.append(finish(i))
            note(f"step {i + 1} of {n_serial} without a first test of the camera rays")
        del os.environ["RTOW_NO_SLAB_TEST"]
        assert plain[-1]["ray_segments"] == serial[-1]["ray_segments"]
        no_slab_ms = sorted(st["trace_ms"] for st in plain)[mid]
    # the older per-ray test (RTOW_CAMERA_SCREEN=slab), and the optional bundle test (RTOW_CAMERA_SCREEN=bundle): a wave's
    # 64 camera rays tested against the table as one packet first.  That is culling per packet, not brute force per
    # ray, so it is never part of `value` or `frac`; the figure is reported for what it is.
    def variant(setting):
        os.environ["RTOW_CAMERA_SCREEN"] = setting
        steps = []
        for i in range(n_serial):
            enqueue(i)
            steps.append(finish(i))
            note(f"step {i + 1} of {n_serial} with RTOW_CAMERA_SCREEN={setting}")
        del os.environ["RTOW_CAMERA_SCREEN"]
        assert steps[-1]["ray_segments"] == serial[-1]["ray_segments"]
        return sorted(st["trace_ms"] for st in steps)[mid]

    bundle_ms = slab_ms = interval_ms = None
    if slab_used and "RTOW_SLAB_AXIS" not in os.environ and "RTOW_CAMERA_SCREEN" not in os.environ:
        if screen == "mask":
            interval_ms = variant("interval")
        if screen in ("mask", "interval"):
            slab_ms = variant("slab")
        bundle_ms = variant("bundle")

    # the frame assembled from all ranks' tiles against the same frame rendered whole on rank 0
    image_verified = None
    if sharded and rank == 0:
        last = gatherers[(n_serial - 1) % depth_in_flight].full          # gathered by the last step
        whole = renderers[0].renderRows(width, height, spp, depth, seed)[0]
        image_verified = bool((last.cpu().numpy() == whole).all())

    if rank == 0:
        flop_per_segment = FLOP_PER_TEST * n_spheres + FLOP_PER_SEGMENT_EXTRA
        avg_trace_s = sum(trace_ms) / len(trace_ms) * 1e-3
        achieved = serial[-1]["ray_segments"] * flop_per_segment / avg_trace_s / 1e12
        paths = width * height * spp
        pass_fraction = screen_pass_fraction(args.workload, spp, screen) if slab_used else 0.0
        if pass_fraction is None or screen not in FLOP_PER_CAMERA_TEST:
            executed = None
        else:
            first = FLOP_PER_CAMERA_TEST[screen] + (2 * screen_pass_fraction(args.workload, spp, "mask_word_pass") if screen == "mask" else 0)
            executed = ((serial[-1]["ray_segments"] - (serial[-1]["paths"] if slab_used else 0)) * FLOP_PER_TEST_EXECUTED
                        + (serial[-1]["paths"] * (first + pass_fraction * FLOP_PER_TEST_EXECUTED) if slab_used else 0)
                        ) * n_spheres / avg_trace_s / 1e12 / PEAK_FP32_VALU_TFLOPS
        out = {
            "metric": "Mrays/sec (ray segments = world.hit queries) book-1 final scene 1200x800 spp500 depth50"
                      + ("" if args.workload == "C2" and not args.spp else f" -- NOT the headline configuration: see config.workload"),
            "value": segments / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / max(1, args.steps) * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",   # every segment after the first; the primary segment is evaluated in binary64
            "data": "synthetic (the reference's randomScene generator from its fixed default RNG state)",
            "config": {
                "workload": f"{args.workload}: book-1 random-spheres scene, {width}x{height}, spp {spp}, depth {depth}, "
                            f"{n_spheres} spheres in a linear list, seed 0xE7",
                "width": width, "height": height, "spp": spp, "max_depth": depth, "n_spheres": n_spheres,
                "parallelism": "single GPU" if not sharded else (
                    f"interleaved {tile}-row tiles over {world} GPUs + RCCL all_gather of 8-bit rows on the render stream"
                    + (" -- REHEARSAL of the collective path with one rank" if args.rehearse_collective else "") if args.backend == "nccl"
                    else f"REHEARSAL: {world} gloo ranks sharing devices (not a benchmark)"),
                "frames_in_flight": depth_in_flight,
                "gathered_image_equals_single_gpu_render": image_verified,
                "latency_ms_per_step": serial_ms,
                "paths_per_step": paths,
                "segments_per_step": segments // max(1, args.steps),
                "mean_segments_per_path": segments / max(1, args.steps) / paths,
            },
            "roofline": {
                "bound": "valu_fp32",
                "kernel": "rtow_trace_kernel",
                "achieved": achieved,
                "peak": PEAK_FP32_VALU_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_VALU_TFLOPS,
                # issue-rate ceilings measured on this chip by bench/ubench_valu.hip (profiles/)
                "measured_peak_v_fma_f32": 126.8,
                "measured_peak_v_pk_fma_f32": 147.7,
                "flop_per_segment": flop_per_segment,
                # `achieved` counts the reference's work (18 FLOP per ray-sphere test, hitrecord.nim:56-61); the
                # kernel screens every sphere with 14 FLOP (7 packed FMAs per pair) and runs the full test on
                # the ~1.2 survivors per ray, so the share of the VALU peak it really issues is lower:
                # (camera-ray segments - one per path - see the first tests - one AND per group of four for the mask test,
                # one subtraction per pair and axis for the interval test on the groups that pass it - and the screen only
                # for the groups that pass both)
                "frac_executed_screen_only": executed,
                "camera_ray_first_test": screen,
                "first_test_pass_fraction": pass_fraction if slab_used else None,
                "note": "achieved = ray segments x the reference's 18 FLOP per ray-sphere pair (+65 per segment); the kernel decides "
                        "most pairs with fewer operations (a quarter of an AND for a camera ray's pair, 7 FMAs otherwise: "
                        "frac_executed_screen_only), so frac measures work delivered against the VALU peak and can approach or "
                        "exceed 1; frac_without_first_test is the same frame with every pair going through the 7-FMA screen",
                "kernel_ms_without_first_test": no_slab_ms,
                "frac_without_first_test": (serial[-1]["ray_segments"] * flop_per_segment / (no_slab_ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS) if no_slab_ms else None,
                # (the names these two figures had before the interval test replaced the slab test as the default)
                "kernel_ms_without_slab_test": no_slab_ms,
                "frac_without_slab_test": (serial[-1]["ray_segments"] * flop_per_segment / (no_slab_ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS) if no_slab_ms else None,
                "kernel_ms_with_interval_test_instead": interval_ms,
                "frac_with_interval_test_instead": None if interval_ms is None else serial[-1]["ray_segments"] * flop_per_segment / (interval_ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS,
                "kernel_ms_with_slab_test_instead": slab_ms,
                "frac_with_slab_test_instead": None if slab_ms is None else serial[-1]["ray_segments"] * flop_per_segment / (slab_ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS,
                # NOT the headline and not brute force: the same frame with RTOW_CAMERA_SCREEN=bundle, where each wave
                # tests the table against its 64 camera rays as one packet before any per-ray test (same image)
                "optional_bundle_test": None if bundle_ms is None else {
                    "kernel_ms": bundle_ms,
                    "frac_on_reference_flop_count": serial[-1]["ray_segments"] * flop_per_segment / (bundle_ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS,
                    "note": "per-packet culling of camera rays; excluded from value and frac"},
                "kernel_ms": avg_trace_s * 1e3,
                # the peak above is 256 CUs x 256 FLOP/clk at the 2.4 GHz boost clock; under this kernel the chip runs
                # slower, and differently from box to box.  Clock read by the kernel itself (shader-clock ticks per
                # tick of the constant 100 MHz counter over the life of its first wave):
                "shader_clock_mhz": clock_mhz,
                "frac_at_measured_clock": (achieved / (PEAK_FP32_VALU_TFLOPS * clock_mhz / 2400.0)) if clock_mhz > 0 else None,
                "traffic": hbm_traffic(args.workload, spp, world),
                "traffic_source": "constant from the committed rocprofv3 PMC passes (profiles/*_traffic.json), not measured in this run",
                "framebuffer_GBps": (n_rows * width * 3) / (sum(total_ms) / len(total_ms) * 1e-3) / 1e9,
            },
            "cpu_baseline": None,
        }
        if world == 1 and not sharded and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(width, height, depth, seed)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())

    for r in renderers:
        r.close()
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
