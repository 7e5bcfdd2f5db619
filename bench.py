#!/usr/bin/env python3
"""Benchmark of the split kd-tree path tracer's render loop on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one render iteration (spath_raytracer_render): one bounce of every
pixel's path = one ray per pixel traced through the kd-tree and shaded.  The
default workload is the million-triangle configuration the headline metric is
quoted on: a 1,310,720-triangle displaced icosphere in a room with an emissive
quad ("c3"), 2048x2048 pixels.  With N > 1 there is one process per GPU: started
by torch.distributed.run (RANK / WORLD_SIZE in the environment), or, when this
script is run plainly with --gpus N, by this script itself, which then starts
that launcher as a child process before anything has touched a GPU and hands
its output and exit code on.  The frame's rows are dealt to the ranks in bands of 8,
each rank renders its rows with the scene replicated, and the resolved rows are
gathered to rank 0 over RCCL at the end of the timed region; total work is
fixed, so the scaling is strong.

Rank 0 prints ONE JSON line: Mrays/s of the whole job, the HBM roofline figure
of the dominant kernel (algorithmic bytes per launch / measured launch time),
a CPU baseline (the oracle restatement on a bounded sample, host cores) and,
where the reference's own kernels have been compiled for gfx950 (oracle/_ref),
their rate on this GPU on a few iterations of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
B_STREAM = 256.0               # per-ray state streaming, bytes (traversal 48 + shading 180 + RNG 8 + per-sample 20)

WORKLOADS = {
    # name: (scene factory args, width, height, description)
    "c3": (("icosphere", 8), 2048, 2048, "1.31M-triangle displaced icosphere + room, 2048x2048, 4 bounces"),
    "c2": (("icosphere", 6), 1024, 1024, "81,920-triangle displaced icosphere + room, 1024x1024, 4 bounces"),
    "c2obj": (("npz", "tests/golden/c2obj_scene.npz"), 1024, 1024,
              "the reference's res/scenes/path_obj3.rsc with mesh/test3.obj (46,966 triangles) in place of the missing "
              "happy.obj, 1024x1024, 4 bounces"),
    "c5": (("soup", 10_000_000), 3840, 2160, "10M-triangle random soup, 3840x2160"),
    "tiny": (("icosphere", 4), 256, 256, "5,120-triangle icosphere + room, 256x256 (smoke)"),
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_scene(spec):
    from raytracer_amd import scenes
    kind, arg = spec
    if kind == "icosphere":
        return scenes.icosphere_scene(arg)
    if kind == "npz":
        return scenes.scene_from_npz(os.path.join(ROOT, arg))
    return scenes.soup_scene(arg)


def host_cpu():
    """(threads to use, description).  The job's share of the host is what the scheduler gives it
    (OMP_NUM_THREADS / the affinity mask), which on a shared GPU host is less than the machine."""
    model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    nproc = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = nproc
    share = affinity
    env = os.environ.get("OMP_NUM_THREADS", "")
    if env.isdigit() and int(env) > 0:
        share = min(share, int(env))
    return share, f"{model}; nproc {nproc}, affinity {affinity}, OMP_NUM_THREADS {env or 'unset'}"


def oracle_tree(sc, r):
    """The oracle's view of the serialised tree the library produced (the reference's wire format)."""
    import oracle
    kdt = r.kdt
    blob = np.ctypeslib.as_array(kdt.buffer, shape=(int(kdt.buffer_size),))
    return oracle.KdTree.from_blob(blob, sc.verts4, sc.idx4, kdt.bounds.min[:3], kdt.bounds.max[:3])


def parity_gates(sc, r, width, full_height, n_local, frame_row0, rows, iterations, threads):
    """The gates a number is reported with, checked AFTER the timed region on the state the device then holds
    (warm-up + timed iterations) against the oracle (CPU restatement): (1) the first hit of EVERY primary ray of
    this rank's pixels -- triangle index and the bits of t, u, v (primary rays and hits are saved by the prepass and
    never change during a render); (2) the accumulated radiance of `rows` frame rows from `frame_row0` (this rank's
    first pixels when they are a band, the middle of the picture otherwise) after the same `iterations` iterations
    with the same RNG streams.  Returns the `parity` object of the bench line."""
    import oracle
    from raytracer_amd.scenes import splitmix32
    tree = oracle_tree(sc, r)
    b = r.spath_buffers()
    oo = r.download(b.origin_o, np.float32, 4 * n_local).reshape(n_local, 4)
    od = r.download(b.origin_d, np.float32, 4 * n_local).reshape(n_local, 4)
    got = r.download(b.source_hits, np.uint32, 4 * n_local).reshape(n_local, 4)
    want = tree.intersect(np.ascontiguousarray(np.concatenate([oo, od], axis=1)), nthreads=threads)
    id_bad = int((got[:, 0] != want["tri"]).sum())
    tuv_bad = int(((got[:, 1] != want["t"].view(np.uint32)) | (got[:, 2] != want["u"].view(np.uint32))
                   | (got[:, 3] != want["v"].view(np.uint32))).sum())
    # radiance of a few rows
    local0 = frame_row0 if n_local == width * full_height else 0          # a band starts at the rank's first pixel
    st = oracle.SpathState(sc, tree, width, rows, frame_row0, full_height)
    st.nthreads = threads
    first = frame_row0 * width
    rb = splitmix32(1, np.arange(first, first + width * rows, dtype=np.uint64))
    rv = splitmix32(2, np.arange(iterations + 1, dtype=np.uint64))
    st.prepass(rb, int(rv[0]))
    for k in range(iterations):
        st.iterate(int(rv[k + 1]))
    base = b.out_tex if isinstance(b.out_tex, int) else b.out_tex.value
    import ctypes
    tex = r.download(ctypes.c_void_p(base + 16 * local0 * width), np.float32, 4 * width * rows).reshape(-1, 4)
    rad = lambda t: t[:, :3] / np.maximum(t[:, 3:4], 1)
    diff = rad(tex).astype(np.float64) - rad(st.out_tex).astype(np.float64)
    return {"against": "oracle (CPU restatement of the reference's kernels), same RNG streams",
            "primary_rays": int(n_local), "first_hit_id_mismatches": id_bad, "first_hit_tuv_bit_mismatches": tuv_bad,
            "primary_rays_that_hit": int((want["t"] != 0).sum()),
            "radiance_rows": [int(frame_row0), int(frame_row0 + rows)], "radiance_iterations": int(iterations),
            "radiance_rmse": float(np.sqrt(np.mean(diff ** 2))), "radiance_max_abs_diff": float(np.abs(diff).max()),
            "radiance_bit_equal": bool((tex.view(np.uint32) == st.out_tex.view(np.uint32)).all()),
            "radiance_rmse_tolerance": 2e-3}


def cpu_baseline(rt, sc, r, width, full_height, threads, cpu_desc, budget_s=12.0, max_iters=96):
    """Oracle (C restatement, OpenMP) on a bounded sample of the same workload:
    the same frame, scene and RNG streams, prepass and the first (null-ray)
    iteration untimed, then whole iterations until about `budget_s` seconds of
    CPU work are spent.  Uses the serialised tree the library produced (the
    reference's wire format), so no second tree build is needed."""
    import oracle
    from raytracer_amd.scenes import splitmix32
    kdt = r.kdt
    blob = np.ctypeslib.as_array(kdt.buffer, shape=(int(kdt.buffer_size),))
    tree = oracle.KdTree.from_blob(blob, sc.verts4, sc.idx4, kdt.bounds.min[:3], kdt.bounds.max[:3])
    st = oracle.SpathState(sc, tree, width, full_height, 0, full_height)
    st.nthreads = threads
    rb = splitmix32(1, np.arange(width * full_height, dtype=np.uint64))
    rv = splitmix32(2, np.arange(max_iters + 2, dtype=np.uint64))
    st.prepass(rb, int(rv[0]))
    st.iterate(int(rv[1]))             # the first iteration traces the wiped (null) rays: untimed
    done = 0
    t0 = time.perf_counter()
    while done < max_iters:
        st.iterate(int(rv[done + 2]))
        done += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            break
    rays = width * full_height * done
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"iterations 2..{done + 1} of the same {width}x{full_height} workload "
                      f"({rays} rays, {dt:.1f} s), oracle/oracle_rt.c with OpenMP on {threads} threads "
                      f"(all of this job's share of the host: {cpu_desc})"}


def reference_kernels_same_gpu(sc, r, workload, width, height, threads, iters=8, groups=8192):
    """The reference's own OpenCL C kernels (compiled from its sources for gfx950 into oracle/_ref/ where the
    reference checkout was mounted, launched by the test harness in oracle/refcl.py) on this GPU, on the same frame,
    scene, tree and RNG streams: a reported baseline like the CPU one, measured after the timed region.  Their
    traversal has an unchecked 16-entry stack, so before each launch the oracle checks the rays and the few that
    would overflow are replaced by null rays.  None where the code objects are not there."""
    from oracle import refcl
    import oracle
    from raytracer_amd.scenes import splitmix32
    tag = {"c3": "ico8", "c2": "ico6"}.get(workload)
    if tag is None or not refcl.available(tag, "default"):
        return None
    kdt = r.kdt
    blob = np.ctypeslib.as_array(kdt.buffer, shape=(int(kdt.buffer_size),)).copy()
    tree = oracle.KdTree.from_blob(blob, sc.verts4, sc.idx4, kdt.bounds.min[:3], kdt.bounds.max[:3])
    rs = refcl.RefScene(refcl.RefProgram(tag, "default"), sc, blob)
    g = refcl.RefSpath(rs, width, height)
    n = width * height
    rb = splitmix32(1, np.arange(n, dtype=np.uint64))
    rv = splitmix32(2, np.arange(iters + 2, dtype=np.uint64))
    g.prepass(rb, int(rv[0]))
    nulled, ms = 0, []
    for k in range(iters + 1):
        rays = g.rays
        deep = tree.ray_stack(rays, nthreads=threads) > 16
        if deep.any():
            rays[deep] = 0
            g.d_rays.write(rays)
            nulled += int(deep.sum())
        x = rs.prog.launch("xorshift_batch", [g.d_random], n)
        t = rs.kd_intersect(g.d_results, g.d_rays, n, groups=groups)
        s_ = rs.prog.launch("segmented_path_trace",
                            [g.d_out_tex, g.d_rays, g.d_origin_rays, g.d_results, g.d_source_results, g.d_progress,
                             g.d_random, rs.materials, rs.meshes, rs.indices, rs.vertices, rs.normals, width,
                             int(rv[k + 1])], n)
        if k > 0:                       # the first iteration traces the wiped (null) rays
            ms.append(x + t + s_)
    per_iter = float(np.mean(ms))
    return {"value": round(n / (per_iter * 1e-3) / 1e6, 1), "unit": "Mrays/s", "kind": "reference kernels, same GPU",
            "ms_per_step": round(per_iter, 3),
            "sample": f"iterations 2..{iters + 1} of the same {width}x{height} workload: xorshift_batch + kdtree_intersection "
                      f"({groups} one-wave groups) + segmented_path_trace, built by ROCm's clang with OpenCL's default options; "
                      f"{nulled} rays that needed more than the reference's 16 stack entries replaced by null rays"}


def gather_stats(rt, sc, r, threads, sample=262_144):
    """Average traversal work per ray of the rays now in the GPU's ray buffer
    (evenly spaced sample), counted by the instrumented oracle."""
    import oracle
    n = r.width * r.height
    b = r.spath_buffers()
    ro = r.download(b.ray_o, np.float32, 4 * n).reshape(n, 4)
    rd = r.download(b.ray_d, np.float32, 4 * n).reshape(n, 4)
    step = max(1, n // sample)
    rays = np.ascontiguousarray(np.concatenate([ro[::step], rd[::step]], axis=1))
    kdt = r.kdt
    blob = np.ctypeslib.as_array(kdt.buffer, shape=(int(kdt.buffer_size),))
    tree = oracle.KdTree.from_blob(blob, sc.verts4, sc.idx4, kdt.bounds.min[:3], kdt.bounds.max[:3])
    _, st = tree.intersect(rays, stats=True, nthreads=threads)
    m = float(st["rays"])
    per_ray = {k: st[k] / m for k in ("nodes_visited", "leaves_visited", "tri_refs", "tri_tests", "pushes")}
    per_ray["max_stack"] = st["max_stack"]
    per_ray["frac_rays_over_16"] = st["rays_over_16"] / m
    b_gather = per_ray["nodes_visited"] * 8 + per_ray["leaves_visited"] * 8 + per_ray["tri_refs"] * 4 + per_ray["tri_tests"] * 36
    return per_ray, b_gather


def spawn_ranks(n):
    """Plain `python bench.py --gpus N`: start N ranks with torch's launcher as a child process and hand on its
    output and exit code.  Nothing in this process has initialised a GPU (counting devices does not)."""
    import socket
    import subprocess
    if not os.environ.get("RT_BENCH_SHARE_GPU"):
        import torch
        have = torch.cuda.device_count()
        if have < n:
            log(f"bench.py: --gpus {n} but {have} GPU(s) are visible")
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default=os.environ.get("RT_BENCH_WORKLOAD", "c3"), choices=sorted(WORKLOADS))
    ap.add_argument("--separate-kernels", action="store_true", help="three launches per iteration, as the reference")
    ap.add_argument("--max-batch", type=int, default=None, help="most iterations folded into one launch")
    ap.add_argument("--no-reorder", action="store_true", help="serve pixels in index order in every batch")
    ap.add_argument("--schedule", type=int, default=None, choices=[0, 1, 2, 3],
                    help="fix the kernel's slot schedule (default: the library's prepass times them)")
    ap.add_argument("--no-run-ahead", action="store_true", help="launches hold only iterations already asked for "
                    "(they then end at every resolve point, every 50 iterations)")
    ap.add_argument("--size", type=int, default=0, help="override the workload's frame: SIZE x SIZE pixels (tuning runs)")
    ap.add_argument("--repeats", type=int, default=1, help="timed regions of --steps iterations each, one after "
                    "another; the median is reported (config.repeats, config.timed_regions_ms)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity gates checked after the timed region")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    # The library reports scene/tree progress on stdout like the reference does;
    # keep stdout for the one JSON line and send everything else to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    n_gpus = world

    # RT_BENCH_FORCE_DIST=1 takes the multi-process path with a single rank, RT_BENCH_BACKEND=gloo and
    # RT_BENCH_SHARE_GPU=1 let several ranks rehearse it on one GPU (RCCL refuses two ranks on one device).
    dist = torch = None
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    if os.environ.get("RT_BENCH_SHARE_GPU"):
        local_rank = 0
    if world > 1 or os.environ.get("RT_BENCH_FORCE_DIST"):
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    if world > 1:
        # every rank builds the same tree: each takes its share of this job's host cores instead of all of them
        # (read by the OpenMP runtime when the library is loaded, below)
        os.environ["OMP_NUM_THREADS"] = str(max(1, host_cpu()[0] // world))
    import raytracer_amd as rt

    spec, width, height, desc = WORKLOADS[args.workload]
    if args.size:
        width = height = args.size
        desc += f" [frame overridden: {width}x{height}]"
    # Interleaved bands of rows: every rank gets rows from all over the frame, so the expensive centre of the
    # picture is shared out evenly (contiguous strips of this scene differ by 2x in cost).
    from raytracer_amd.strips import BAND_ROWS, band_frame_index, band_partition, band_share_sizes
    try:
        first, band, stride, strip = band_partition(height, world, rank)
    except ValueError as e:
        raise SystemExit(str(e))

    t0 = time.perf_counter()
    sc = make_scene(spec)
    t_scene = time.perf_counter() - t0
    r = rt.Renderer(sc, width, strip, device=local_rank, tile_y0=first, full_height=height,
                    bands=(band, stride) if world > 1 else None)
    t0 = time.perf_counter()
    r.build()
    t_build = time.perf_counter() - t0
    r.init_spath(seed=1, separate_kernels=args.separate_kernels, resolve_interval=50, quiet=True,
                 max_batch=args.max_batch, reorder=not args.no_reorder, schedule=args.schedule,
                 run_ahead=not args.no_run_ahead)
    # the prepass ends with the library's calibration of the kernel's slot schedules (untimed, like the tree build)
    r.spath_prepass(num_iterations=1 << 30)
    if rank == 0:
        log(f"[bench] {desc}: {sc.num_tris} triangles, scene {t_scene:.1f}s, build {t_build:.1f}s, "
            f"{r.kdt.num_nodes_total} nodes, {width}x{strip} pixels per rank, device {r.device_props()['name']}")

    def sync_all():
        r.sync()
        if torch is not None:
            torch.cuda.synchronize()
            dist.barrier()

    if torch is not None and args.schedule is None and not args.separate_kernels:
        # Every rank timed the kernel's slot schedules on its own share in its prepass; the shares are
        # interleaved bands of one picture, so they should agree, and a rank that a noisy timing sent the other
        # way would set the job's time.  All ranks take the choice most of them made (results do not depend on it).
        mine = torch.zeros(4, dtype=torch.int64, device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
        mine[int(r.spath.contents.schedule)] = 1
        dist.all_reduce(mine)
        r.spath.contents.schedule = int(torch.argmax(mine).item())
        r.spath.contents.schedule_fixed = 1
    n_local = width * strip
    frame = wire = gathered = None
    shares = band_share_sizes(width, height, world) if world > 1 else [n_local]
    pad = max(shares)
    if torch is not None:
        # one preallocated [world, pad] tensor on rank 0 takes the gather; shares are padded to the largest
        frame = torch.zeros(pad, dtype=torch.int32, device=f"cuda:{local_rank}")
        wire = frame if backend == "nccl" else torch.zeros(pad, dtype=torch.int32)    # gloo moves host tensors
        gathered = torch.empty((world, pad), dtype=torch.int32, device=wire.device) if rank == 0 else None

    def gather_frame():
        # resolve this rank's rows, hand them to torch and gather the frame on rank 0 (RCCL)
        r.L.rt_k_resolve_avg(r.dev, r.rctx.contents.cl_output_buffer, r.spath.contents.cl_path_output_buffer, n_local)
        r.L.rt_mem_copy(r.dev, frame.data_ptr(), r.rctx.contents.cl_output_buffer, n_local * 4)
        r.sync()
        t_copied = time.perf_counter()
        if wire is not frame:
            wire.copy_(frame)
        dist.gather(wire, list(gathered.unbind(0)) if rank == 0 else None, dst=0)
        return t_copied

    # Renderer.spath_render ends the render where its calls end while they run, so the library's launches never
    # reach past the warm-up into the timed iterations, nor past those.  Nothing but the barrier lies between the
    # warm-up launch and the timed one: the parity gates (seconds of CPU work) are checked after the timed region,
    # so the timed launch starts on a GPU that has just been busy, at the clocks a render runs at.
    r.spath_render(args.warmup)
    if torch is not None:
        gather_frame()      # part of the warm-up: the first gather sets up the point-to-point connections it uses
    sync_all()
    r.timing(True)
    r.timing_reset()

    regions, split = [], None
    for rep in range(args.repeats):
        sync_all()
        t0 = time.perf_counter()
        r.spath_render(args.steps)
        if torch is not None:
            r.sync(); ta = time.perf_counter()                  # render (this rank's launch has ended)
            tb = gather_frame(); tc = time.perf_counter()       # resolve + copy | gather
            sync_all(); td = time.perf_counter()                # barrier: the job ends when its slowest rank does
            split = [1e3 * (ta - t0), 1e3 * (tb - ta), 1e3 * (tc - tb), 1e3 * (td - tc)]
        else:
            sync_all()
        regions.append(time.perf_counter() - t0)
    if torch is not None:
        t = torch.tensor(regions + split, dtype=torch.float64, device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
        every = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        every = torch.stack(every).cpu().numpy()
        regions = [float(x) for x in every[:, :args.repeats].max(axis=0)]       # MAX over ranks, per timed region
        rank_split = every[:, args.repeats:]
    else:
        rank_split = None
    dt = float(np.median(regions))
    done_iterations = args.warmup + args.repeats * args.steps

    # dominant kernel: the multi-iteration launch (traversal + shading of a batch of
    # iterations), or the traversal launch when iterations are launched one by one
    kname = "kd_intersect" if args.separate_kernels else "spath_run"
    launches, kms = r.timing_get(kname)
    _, shade_ms = r.timing_get("spath_trace")
    r.timing(False)

    # The image after warmup + steps iterations does not depend on how many GPUs rendered it:
    # its checksum goes into the line so runs at different N can be compared.
    import zlib
    if torch is not None:
        if rank == 0:       # the ranks' bands back into frame order
            whole = gathered.reshape(-1)[band_frame_index(width, height, world).to(gathered.device)]
            frame_crc = zlib.crc32(whole.cpu().numpy().tobytes())
        else:
            frame_crc = 0
    else:
        frame_crc = zlib.crc32(r.spath_resolve().tobytes())

    parity, parity_ok = None, None
    if rank == 0 and not args.no_parity:
        single = world == 1
        rows = min(16 if single else BAND_ROWS, strip)
        try:
            parity = parity_gates(sc, r, width, height, n_local, (height // 2 - rows // 2) if single else 0, rows,
                                  done_iterations, args.cpu_threads or host_cpu()[0])
            parity_ok = (parity["first_hit_id_mismatches"] == 0 and parity["first_hit_tuv_bit_mismatches"] == 0
                         and parity["radiance_rmse"] <= parity["radiance_rmse_tolerance"])
        except Exception as e:              # a checker that could not run is reported as such; the line says so
            log(f"[bench] parity gates not checked: {e!r}")
            parity = {"error": repr(e)}

    exit_code = 0
    if rank == 0:
        total_rays = width * height * args.steps
        value = total_rays / dt / 1e6
        threads, cpu_desc = host_cpu()
        threads = args.cpu_threads or threads
        per_ray, b_gather = gather_stats(rt, sc, r, threads)
        b_ray = B_STREAM + b_gather
        avg_ms = kms / max(launches, 1)
        iters_per_launch = args.repeats * args.steps / max(launches, 1)      # 1 with --separate-kernels
        rays_per_launch = n_local * iters_per_launch
        achieved = (b_ray * rays_per_launch) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM bytes per launch and what limits the kernel, from the stored rocprofv3 --pmc passes of this workload
        # (profiles/pmc_passes.sh; a profiler cannot run inside this process): scaled by this run's iterations
        traffic = counter_frac = None
        limiter = {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                rec = json.load(open(tpath))[f"{args.workload}:{kname}"]
                traffic = round(rec["hbm_bytes_per_iteration"] * iters_per_launch * n_local / (width * height))
                counter_frac = round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if avg_ms > 0 else None
                limiter = {k: rec[k] for k in ("limiter", "valu_insts_per_ray", "useful_lane_frac", "profile") if k in rec}
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/sec (primary+bounce)", "value": round(value, 2), "unit": "Mrays/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32",
            "data": "reference mesh files (frozen in tests/golden/c2obj_scene.npz), synthetic RNG streams" if spec[0] == "npz" else "synthetic",
            "parity_ok": parity_ok,
            "config": {"workload": f"{args.workload}: {desc}", "triangles": int(sc.num_tris), "width": width,
                       "height": height, "rows_per_gpu": strip, "kernel": kname,
                       "schedule": int(r.spath.contents.schedule),
                       "schedule_calibration_ms_per_iteration": [round(float(x), 4) for x in r.spath.contents.trial_ms]
                       if int(r.spath.contents.trial_state) == 2 else None,
                       "run_ahead": not args.no_run_ahead, "frame_crc32": f"{frame_crc:08x}",
                       "frame_after_iterations": done_iterations,
                       "repeats": args.repeats,
                       "timed_regions_ms": [round(1e3 * x, 3) for x in regions],
                       "partition": f"{BAND_ROWS}-row bands dealt to {world} ranks ({min(shares) // width}-{max(shares) // width} rows "
                                    f"each) + RCCL gather of the resolved frame" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": "profiles/traffic.json: FETCH_SIZE + WRITE_SIZE of rocprofv3 --pmc passes over this "
                                           "workload's launch, scaled to this run's iterations per launch -- not measured in this run",
                         "counter_frac": counter_frac, **limiter,
                         "kernel": kname, "avg_launch_ms": round(avg_ms, 4), "launches": launches,
                         "bytes_per_ray": round(b_ray, 1), "rays_per_launch": int(rays_per_launch),
                         "iterations_per_launch": round(iters_per_launch, 2),
                         "shade_avg_launch_ms": round(shade_ms / max(launches, 1), 4),
                         "per_ray": {k: round(v, 3) for k, v in per_ray.items()}},
        }
        out["parity"] = parity
        # the target render (256 samples x 4 bounces = 1024 iterations) against the reference's own kernels on this
        # GPU model: a stored, dated figure of tests/tools/ref_kernels_parity.py (minutes of run time), quoted here
        import glob
        found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"{args.workload}_full_size_parity_1024_iterations.json")))
        quoted = found[-1] if found else ""
        if isinstance(parity, dict) and quoted:
            try:
                q = json.load(open(quoted))
                parity["quoted_full_render_vs_reference_kernels"] = {
                    "source": os.path.relpath(quoted, ROOT), "measured": "by tests/tools/ref_kernels_parity.py (minutes of run time), not in this run",
                    "iterations": q["iterations"], "radiance_rmse": q["radiance_rmse"],
                    "pixels_left_out_reference_stack_overflow": q["pixels_left_out"],
                    "pixels_within_1e-3": q["pixels_within_1e-3"]}
            except Exception:
                pass
        if rank_split is not None:
            # where each rank's share of the last timed region went (host clock, ms)
            out["ranks"] = [{"rank": k, "rows": shares[k] // width, "render_ms": round(float(v[0]), 3),
                             "resolve_copy_ms": round(float(v[1]), 3), "gather_ms": round(float(v[2]), 3),
                             "barrier_ms": round(float(v[3]), 3)} for k, v in enumerate(rank_split)]
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(rt, sc, r, width, height, threads, cpu_desc)
            out["cpu_baseline"]["value"] = round(out["cpu_baseline"]["value"], 3)
            try:
                ref = reference_kernels_same_gpu(sc, r, args.workload, width, height, threads)
            except Exception as e:          # the harness' launcher must not cost the benchmark its line
                log(f"[bench] reference kernels not timed: {e}")
                ref = None
            if ref is not None:
                out["reference_gpu"] = ref
        if parity_ok is False:
            # a gate that ran and failed: the number does not stand for the reference's computation
            log(f"[bench] PARITY GATE FAILED: {parity}")
            out["value_unchecked"] = out["value"]
            out["value"] = None
            exit_code = 3
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    if torch is not None:
        dist.barrier()
        dist.destroy_process_group()
    r.close()
    if exit_code:
        sys.exit(exit_code)


if __name__ == "__main__":
    main()
