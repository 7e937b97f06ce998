#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X path-tracing back end.

Metric (BASELINE.json): Mrays/s (primary + secondary) and ms/frame, 1280x720, 64 spp path trace, bunny.obj BVH-SAH
(`configs[1]`).  A *step* = one pass of the hot path over one batch of synthetic input = clear the accumulator and
render 64 frames (spp 1..64, passes = 1, depthLimit 5) of the bunny scene at 1280x720, inputs (BVH, triangles,
textures) already resident in HBM.  Rays = FindNearest calls (primary + secondary), counted on the device.

    python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run, one rank per GPU)

Steps are the consecutive 64-frame windows of ONE progressive render (spp 1..64, 65..128, ...): they are independent
((tile, frame) streams, renderer.cpp:120) except for the accumulation order, so the K steps are submitted as one
crt_render of 64*K frames: the back end covers up to 64 windows with ONE render_pool_kernel grid (a wavefront per
(tile, 128 consecutive frames), expensive tiles first) and adds the samples to the accumulator in frame order behind it.
`value` is therefore PIPELINED throughput; the line also carries `single_render` = ONE 1280x720 / 64-spp render on its own
(submit, wait), which ends on its most expensive tile and is several times slower per step.
Multi-GPU (default `--split tiles`, BASELINE config 5 / SURVEY 8(e)): the 16x16 tiles of the ONE image are dealt round-robin
over the ranks (tile ownership: every pixel is non-zero on exactly one rank) and ONE RCCL reduce of the float4 accumulator
over xGMI closes the job: the image is bit-identical to the single-GPU one (strong scaling: the K windows are shared out).
`--split frames` (weak scaling): every rank renders its own K windows of the full image, image = 64*K*N spp.
`value` = rays of all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line with the contract fields + "roofline" (dominant kernel = the render kernel, HIP events
on the launch stream) + "cpu_baseline" (the CPU oracle on this box's host cores, bounded sample, rank 0, N = 1 only).
"""
import argparse
import importlib.util
import json
import os
import sys
import time

import numpy as np

# ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels that share a queue serialise; the
# back end overlaps independent 64-frame launches on several streams, so give the runtime enough queues BEFORE it initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

REPO = os.path.dirname(os.path.abspath(__file__))
ASSETS = os.path.join(REPO, "assets")
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
VALU_ISSUE_PEAK_GINSTR = 1050.0   # measured fp32 VALU issue peak of the chip, G wave-instructions/s (tools/microbench/valu_peak.hip, profiles/r03_valu_peak.txt)
VALU_ISSUE_PEAK_THEORETICAL = 1229.0   # 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 fp32 instruction (MI355X_MICROARCH.md)
VALU_INT_PEAK_GINSTR = 605.0      # measured rate of integer / compare VALU instructions (v_lshl, v_xor, v_cmp + v_cndmask pairs): half the fp32 rate (same microbench)


def load_crt():
    spec = importlib.util.spec_from_file_location("cpu_ray_tracer_amd", os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["cpu_ray_tracer_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def algorithmic_bytes(c):
    """64 B per interior iteration (both children), 40 B per triangle test, 64 B per BLAS visit,
    76 B per mesh hit, 32 B per primary sample (float4 accumulate read + write)."""
    return 64 * c["interior_iters"] + 40 * c["tri_tests"] + 64 * c["blas_visits"] + 76 * c["mesh_hits"] + 32 * c["primary"]


def cpu_baseline(scene_xml, kind, W, H, budget_s=float(os.environ.get("CRT_BENCH_CPU_BUDGET_S", "15"))):
    """Times the CPU oracle (kind "port": the repo's restatement of the reference algorithm) on this box's cores."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import orc
    threads = os.cpu_count() or 1
    try:
        threads = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:                                   # container CPU quota (cgroup v2 "max period"): more threads than that only get throttled
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            threads = max(1, min(threads, int(int(quota) / int(period))))
    except Exception:
        pass
    o, _ = orc.load_scene(scene_xml, kind, ASSETS)
    o.renderer_init(W, H)
    o.render(1, threads)                       # warm-up frame (spp 1), also sizes the sample
    o.reset_counters()
    t0 = time.perf_counter()
    o.render(1, threads)
    one = time.perf_counter() - t0
    frames = int(max(1, min(600, budget_s / max(one, 1e-3) - 1)))      # about 15 s of CPU work
    t1 = time.perf_counter()
    o.render(frames, threads)
    dt = (time.perf_counter() - t1) + one
    c = o.counters()
    # SURVEY 8(d) also asks for the single-thread figure: two more frames on one thread
    o.reset_counters()
    t2 = time.perf_counter()
    o.render(2, 1)
    one_thread = o.counters()["rays"] / (time.perf_counter() - t2) / 1e6
    return {"value": round(c["rays"] / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port", "value_1_thread": round(one_thread, 3),
            "scaling_efficiency": round(c["rays"] / dt / 1e6 / max(one_thread * threads, 1e-9), 3),
            "sample": "%d frames (spp 2..%d) of the same %dx%d scene, %d threads (= usable host cores: affinity capped by the cgroup CPU quota), %.1f s" % (frames + 1, frames + 2, W, H, threads, dt),
            "ms_per_frame": round(dt / (frames + 1) * 1e3, 2)}


def workload_text(scene, kind, W, H, spp, steps, launches, world, split, coll_lib, collective):
    """config.workload of the output line (a plain function so that the N > 1 wording is testable without a GPU)"""
    text = ("%s %s BVH-SAH path tracer, %dx%d, %d spp/step (passes=1, depthLimit=5); 1 step = %d frames = the next spp window of a "
            "progressive render; the %d steps are one crt_render job (%d render kernel launch(es), a wavefront per (tile, 128 consecutive frames), "
            "+ ordered accumulate), one sync at the end" % (scene, "TLASFileScene" if kind else "FileScene", W, H, spp, spp, steps, launches))
    if world > 1 and split == "frames":
        text += "; every one of the %d ranks renders its own %d windows, ONE %s all-reduce of the float4 accumulator closes the job" % (world, steps, coll_lib)
    elif world > 1:
        text += "; the image's tiles are dealt round-robin over %d ranks (tile ownership), ONE %s %s of the float4 accumulator closes the job" % (world, coll_lib, collective)
    return text


def spawn_command(gpus, argv, port=None):
    """`bench.py --gpus N` without an external launcher: the command that starts the N ranks (one per GPU) on this node"""
    if port is None:
        port = 29500 + (os.getpid() % 2000)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def choose_collective(dist, torch, device, wanted):
    """The collective that closes a tile-split job, decided ONCE and identically on every rank before the warm-up: `reduce` is probed on a tiny tensor
    (a backend may refuse it for device tensors — gloo rehearsals), the ranks agree on the outcome with an all_reduce(MIN) of a success flag."""
    if wanted != "reduce":
        return wanted
    ok = 1
    try:
        probe = torch.ones(4, dtype=torch.float32, device=device)
        dist.reduce(probe, dst=0, op=dist.ReduceOp.SUM)
        if str(device) != "cpu":
            torch.cuda.synchronize()
    except RuntimeError:
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return "reduce" if int(flag.item()) == 1 else "all_reduce"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--scene", default="bunny_scene.xml")
    ap.add_argument("--kind", type=int, default=0, help="0 = FileScene (single BVH), 1 = TLASFileScene")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=7, help="HIP streams the render launches rotate over (crt_config.renderStreams)")
    ap.add_argument("--split", choices=["frames", "tiles"], default="tiles",
                    help="N > 1: 'tiles' (default, strong scaling, north_star / BASELINE config 5) = the K windows of ONE image with the 16x16 tiles dealt "
                         "round-robin over the ranks, exact image; 'frames' (weak scaling) = every rank renders its own K windows of the full image")
    ap.add_argument("--reduce", choices=["reduce", "all_reduce"], default="reduce",
                    help="N > 1: the collective that closes the job: ncclReduce of the float4 accumulator to rank 0 (default; SURVEY 8(e)) or ncclAllReduce")
    ap.add_argument("--no-single-render", action="store_true", help="skip the single-render latency measurement (three 64-frame renders on their own after the job)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no external launcher: start the N ranks ourselves — before anything in this process touches the GPU (no torch import yet) — and hand their exit code on;
        # rank 0 of the children prints the one JSON line
        import subprocess
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call(spawn_command(args.gpus, sys.argv[1:])))

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit("bench.py --gpus %d was started with WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    if world > 1 or os.environ.get("CRT_BENCH_FORCE_DIST"):      # (FORCE_DIST: a one-rank RCCL rehearsal on a one-GPU box — init, reduce / all_reduce, barrier all run)
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("CRT_BENCH_BACKEND", "nccl")      # "gloo" only to rehearse the N > 1 code path on a one-GPU box
        local_rank %= max(torch.cuda.device_count(), 1)            # (device_count does not initialise the GPU)
        if backend == "nccl":
            if not torch.cuda.is_available():
                sys.exit("bench.py needs an MI355X: there is no CPU path for the product")
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    if not torch.cuda.is_available():
        if dist is not None:
            dist.barrier(); dist.destroy_process_group()
        sys.exit("bench.py needs an MI355X: there is no CPU path for the product (rank %d of %d)" % (rank, world))
    device = local_rank if world > 1 else 0
    torch.cuda.set_device(device)

    crt = load_crt()
    W, H, SPP = args.width, args.height, args.spp
    xml = os.path.join(ASSETS, "scenes", args.scene)
    scene = crt.HostScene(xml, args.kind, ASSETS)                  # XML + OBJ + textures + SAH-BVH build on the CPU
    tiles = (W // 16) * (H // 16)
    tsplit = crt.tile_partition(rank, world, tiles) if (world > 1 and args.split == "tiles") else (0, 1, -1)
    ctx = crt.Context(W, H, device=device, render_streams=args.streams, tile_first=tsplit[0], tile_stride=tsplit[1], tile_count=tsplit[2])
    scene.upload(ctx)                                             # one-time flatten + copy to HBM
    acc = torch.zeros(H, W, 4, dtype=torch.float32, device="cuda:%d" % device)
    ctx.bind_accumulator(acc.data_ptr())
    def window(i, n_steps):
        """spp counter of the first frame of this rank's i-th step of an n_steps job: steps are consecutive 64-frame windows of ONE
        progressive render; rank r owns the windows r*n_steps .. r*n_steps + n_steps - 1"""
        return crt.spp_window((rank * n_steps if args.split == "frames" else 0) + i, SPP)

    # one counted pass over the same windows with a statistics context (untimed) -> algorithmic bytes per launch
    sctx = crt.Context(W, H, device=device, collect_stats=True, tile_first=tsplit[0], tile_stride=tsplit[1], tile_count=tsplit[2])
    scene.upload(sctx)
    sctx.render(window(0, args.steps), SPP * args.steps, 1)      # (a statistics context renders window by window)
    sctx.sync()
    counts = {k: v / args.steps for k, v in sctx.counters().items()}
    sctx.close()

    collective = {"used": args.reduce if dist is None else choose_collective(dist, torch, "cuda:%d" % device, args.reduce)}

    def run(n_steps):
        """n_steps steps = 64*n_steps frames submitted as one job: the back end renders up to 64 windows per render_tiles_kernel
        launch and accumulates them in frame order; one sync (and, for N > 1, ONE RCCL all-reduce of the float4 accumulators
        over xGMI) closes the job."""
        ctx.clear()
        ctx.render(window(0, n_steps), SPP * n_steps, 1)
        ctx.sync()
        if dist is not None:
            if collective["used"] == "reduce":
                crt.reduce_accumulator(acc, dist, 0)
            else:
                crt.allreduce_accumulator(acc, dist)
            torch.cuda.synchronize()

    ctx.reserve(SPP * max(args.steps, args.warmup), 1)             # sample-slab pool for the whole job, allocated outside the timed region
    run(args.warmup)
    tm_warm = ctx.timing()
    ctx.reset_counters()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tm = ctx.timing()
    kernel_ms, acc_ms, launches = tm["render_kernel_ms"], tm["resolve_kernel_ms"], tm["render_launches"]
    rays = ctx.counters()["rays"]
    # latency of ONE step on its own (submit, wait), next to the job throughput: the literal "one WxH / SPP-spp render"
    lat = []
    if not args.no_single_render:
        for i in range(16):                         # (the back end tunes its latency mode over the first 7 single-window launches: probe -> solved block tables, confirms and keeps the fastest)
            ctx.clear(); ctx.sync()
            t1 = time.perf_counter()
            ctx.render(1, SPP, 1)
            ctx.sync()
            lat.append((time.perf_counter() - t1) * 1e3)
        ctx.timing()

    if dist is not None:
        t = torch.tensor([elapsed, float(rays)], dtype=torch.float64, device="cuda:%d" % device)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0].item())
        rays = int(t[1].item())
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_step = elapsed / args.steps * 1e3
    coll_lib = "RCCL" if (dist is None or dist.get_backend() == "nccl") else dist.get_backend()          # (gloo only in one-GPU rehearsals of the N > 1 path)
    # Dominant kernel = render_tiles_kernel.  Every launch of it in this process (the warm-up job and the timed job) enters the
    # average, so that avg_launch_ms is the figure rocprofv3 --kernel-trace --stats reports for the same command; bytes = SURVEY 8(d)'s
    # per-ray / per-sample figures x the device counters of one step x the steps a launch covers.
    # (the back end picks render_pool_kernel for launches many times larger than the machine and render_tiles_kernel below that: the warm-up job
    # enters the average only when it ran the same kernel as the timed job)
    job_kernel = "render_pool_kernel" if tm.get("pool_launches", 0) == launches and launches > 0 else "render_tiles_kernel"
    warm_same = (tm_warm.get("pool_launches", 0) == tm_warm["render_launches"]) == (job_kernel == "render_pool_kernel")
    all_launches = launches + (tm_warm["render_launches"] if warm_same else 0)
    all_ms = kernel_ms + (tm_warm["render_kernel_ms"] if warm_same else 0.0)
    avg_launch_ms = all_ms / max(all_launches, 1)
    launches_per_step = launches / args.steps
    alg_bytes_launch = algorithmic_bytes(counts) * (args.steps + (args.warmup if warm_same else 0)) / max(all_launches, 1)
    achieved = alg_bytes_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    job_launch_ms = kernel_ms / max(launches, 1)
    # HBM bytes and VALU wave-instructions of the timed job's render kernels (render_pool_kernel and, in a split job, the block-table render_tiles_kernel beside it)
    # from rocprofv3 PMC passes of THIS command (tools/profile_job.sh -> profiles/job_counters.json, keyed by job shape); a shape that has not been profiled
    # falls back to the per-window figures of the 64-window pool-only job (profiles/hbm_traffic.json) and says so in counters_from
    job_shape = "%s|%d|%dx%d|spp%d|steps%d|warmup%d|gpus%d" % (args.scene, args.kind, W, H, SPP, args.steps, args.warmup, world)
    traffic = valu_instrs = lane_util = l2_hit = None
    counters_from = None
    windows_per_launch = (1 / launches_per_step) if launches_per_step else 0
    try:
        jc = json.load(open(os.path.join(REPO, "profiles", "job_counters.json"))).get(job_shape)
    except Exception:
        jc = None
    if jc and "fetch_bytes_raw" in jc and "valu_wave_instructions" in jc:
        traffic = int((jc["fetch_bytes_raw"] + jc["write_bytes"]) / max(launches, 1))
        valu_instrs = int(jc["valu_wave_instructions"] / max(launches, 1))
        lane_util = jc.get("lane_utilisation"); l2_hit = jc.get("l2_hit_rate")
        counters_from = "%s: rocprofv3 PMC passes of this exact command (job shape %s), every render kernel of the timed job" % (jc.get("file", "profiles/job_counters.json"), job_shape)
    else:
        pmc_path = os.path.join(REPO, "profiles", "hbm_traffic.json")
        try:
            t = json.load(open(pmc_path))
            if t.get("workload") == [args.scene, args.kind, W, H, SPP] and world == 1 and job_kernel in t.get("kernel", ""):
                pw = t["per_window"]
                traffic = int((pw["fetch_bytes_raw"] + pw["write_bytes"]) * windows_per_launch)
                valu_instrs = int(pw["valu_wave_instructions"] * windows_per_launch)
                lane_util = t.get("lane_utilisation"); l2_hit = t.get("l2_hit_rate")
                counters_from = "profiles/hbm_traffic.json: PMC of the 64-window pool-only job, per window x %.1f windows of this launch (this job shape has not been profiled: a model, not a counter)" % windows_per_launch
        except Exception:
            pass
    single_ms = sorted(lat[-5:])[2] if lat else None
    out = {
        "metric": "Mrays/sec (primary+secondary), %dx%d %dspp path trace — pipelined throughput over %d consecutive %d-spp windows of one progressive render (single_render = one such render alone)" % (W, H, SPP, args.steps, SPP),
        "value": round(rays / elapsed / 1e6, 2),
        "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 4),
        "ms_per_frame": round(ms_step / SPP, 5),
        "higher_is_better": True, "scaling": "weak" if args.split == "frames" else "strong", "vs_baseline": None, "job_shape": job_shape,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload_text(args.scene, args.kind, W, H, SPP, args.steps, launches, world, args.split, coll_lib, collective["used"]),
                   "latency_ms_single_step": round(single_ms, 3) if single_ms else None,
                   "collective": None if dist is None else collective["used"], "rccl_ranks": None if dist is None else dist.get_world_size(), "backend": None if dist is None else dist.get_backend(),
                   "rays_per_step_rank0": round(counts["rays"]), "rays_per_primary": round(counts["rays"] / max(counts["primary"], 1), 4),
                   "triangles": scene.triangle_count(), "parallelism": "tile-wave x%d" % world},
        "single_render": None if not single_ms else {"ms": round(single_ms, 3), "mrays_s": round(counts["rays"] / single_ms / 1e3, 1), "first_ms": round(lat[0], 3), "second_ms": round(lat[1], 3),
                                                     "what": "ONE %dx%d / %d-spp render on an idle GPU: clear, crt_render(1, %d, 1), sync (16 renders; `first_ms` = the first render after a camera / scene change, cost probe included: its block table is solved from the probe's 512 paths per tile; `second_ms` = the table solved from the tile costs the first render measured; renders 2-7 are the latency mode's stages and confirmations, `ms` = median of the last 5)" % (W, H, SPP, SPP)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "hbm_actual_frac": None if not traffic or job_launch_ms <= 0 else round(traffic / (job_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                     "useful_lane_issue_frac": None if not valu_instrs or not lane_util or job_launch_ms <= 0 else round(valu_instrs / (job_launch_ms * 1e-3) / 1e9 / VALU_ISSUE_PEAK_GINSTR * lane_util, 4),
                     "limiter": "instruction issue on divergent fp32 / integer code, NOT HBM: `bound`/`frac` are SURVEY 8(d)'s nominal yardstick (algorithmic bytes against the HBM peak) and reach or exceed 1 only "
                                "because the 0.75 MB scene is served by L2 (see l2_hit_rate) — the DRAM traffic is hbm_actual_frac of the peak; what grades the kernel is useful_lane_issue_frac = VALU issue rate / issue peak x lane utilisation",
                     "counters_from": counters_from, "l2_hit_rate": None if l2_hit is None else round(l2_hit, 4),
                     "kernel": job_kernel, "job_split": bool(tm.get("split_launches", 0)), "avg_launch_ms": round(avg_launch_ms, 4), "launches": all_launches,
                     "algorithmic_bytes_per_launch": int(alg_bytes_launch),
                     "job_launch_ms": round(job_launch_ms, 4), "job_launch_windows": round(1 / launches_per_step, 2) if launches_per_step else None,
                     "job_launch_achieved": round(algorithmic_bytes(counts) / max(launches_per_step, 1e-9) / (job_launch_ms * 1e-3) / 1e9, 2) if job_launch_ms > 0 else None,
                     "job_achieved": round(rays / elapsed * (algorithmic_bytes(counts) / max(counts["rays"], 1)) / 1e9, 2),
                     "accumulate_kernel_ms_per_step": round(acc_ms / args.steps, 4),
                     "valu_issue": None if not valu_instrs or job_launch_ms <= 0 else {
                         "wave_instructions_per_job_launch": valu_instrs, "achieved_ginstr_s": round(valu_instrs / (job_launch_ms * 1e-3) / 1e9, 1),
                         "peak_ginstr_s": VALU_ISSUE_PEAK_GINSTR, "peak_theoretical_ginstr_s": VALU_ISSUE_PEAK_THEORETICAL, "peak_int_cmp_ginstr_s": VALU_INT_PEAK_GINSTR,
                         "frac": round(valu_instrs / (job_launch_ms * 1e-3) / 1e9 / VALU_ISSUE_PEAK_GINSTR, 4),
                         "frac_of_theoretical": round(valu_instrs / (job_launch_ms * 1e-3) / 1e9 / VALU_ISSUE_PEAK_THEORETICAL, 4),
                         "valu_wave_instructions_per_ray": round(valu_instrs / max(counts["rays"] * windows_per_launch, 1), 3),
                         "lane_utilisation": None if lane_util is None else round(lane_util, 4),
                         "note": "SQ_INSTS_VALU of the job's render kernels / the launch's duration; peaks: measured fp32 mul / add / fma / cndmask issue peak of the chip (tools/microbench/valu_peak.hip, profiles/r03_valu_peak.txt: 1 050 G wave-instr/s), "
                                 "theoretical 256 CU x 4 SIMD x 2.4 GHz / 2 cycles = 1 229 G/s, and the measured rate of integer / compare instructions (shifts, xor, v_cmp: 605 G/s — the RNG and the predicate logic run at half rate); "
                                 "lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)"},
                     "counters_per_step": {k: round(v) for k, v in counts.items()},
                     "note": "achieved = mean algorithmic bytes per " + job_kernel + " launch / mean launch duration over all its launches in this process (warm-up job + timed job; HIP events on the launch stream) = the average rocprofv3 --stats reports; job_launch_* = the timed job's launch alone (job_split: its most expensive tiles ran beside it in a concurrent render_tiles_kernel launch driven by a block table, inside the same event pair); job_achieved = algorithmic GB/s over the whole timed region incl. the ordered accumulate; the bytes are algorithmic (SURVEY 8(d)) and mostly served by L2 — traffic = HBM bytes (FETCH_SIZE + WRITE_SIZE, x 1024) of the job's render kernels per launch from the PMC passes named in counters_from, hbm_actual_frac = traffic / job launch time / peak"},
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(xml, args.kind, W, H)
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
