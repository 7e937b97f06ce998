#!/usr/bin/env python3
"""Diagnostic (-DCRT_POOL_DENS build, CRT_LIB_PATH=...): how often every section of render_pool_kernel's loop runs and with how many of
its 64 lanes, for one pool-only job (default 32 windows of the bunny scene at 1280x720).  Prints one JSON line.
    python tools/pool_density.py [scene.xml kind [windows [W H]]]"""
import ctypes as C, importlib.util, json, os, sys, time
os.environ.setdefault("CRT_ENABLE_DEBUG_HOOKS", "1")      # the library reads its diagnostic environment switches only for processes that opt in
os.environ.setdefault("CRT_RENDER_KERNEL", "pool_always")
os.environ.setdefault("CRT_SPLIT_OFF", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("cpu_ray_tracer_amd", os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
crt = importlib.util.module_from_spec(spec); spec.loader.exec_module(crt)
A = os.path.join(REPO, "assets")
xml = sys.argv[1] if len(sys.argv) > 1 else "bunny_scene.xml"; kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
K = int(sys.argv[3]) if len(sys.argv) > 3 else 32
W = int(sys.argv[4]) if len(sys.argv) > 4 else 1280; H = int(sys.argv[5]) if len(sys.argv) > 5 else 720
sc = crt.HostScene(os.path.join(A, "scenes", xml), kind, A)
ctx = crt.Context(W, H); sc.upload(ctx); ctx.reserve(64 * K, 1)
ctx.render(1, 64 * K, 1); ctx.sync(); ctx.timing()
L = crt.lib()
buf = (C.c_uint64 * 32)()
has = hasattr(L, "crt_debug_pool_density")
if has: L.crt_debug_pool_density(buf, 1)
ctx.reset_counters(); ctx.clear(); ctx.sync()
t0 = time.perf_counter(); ctx.render(1, 64 * K, 1); ctx.sync(); wall = (time.perf_counter() - t0) * 1e3
tm = ctx.timing(); c = ctx.counters()
out = {"scene": xml, "windows": K, "job_ms": round(wall, 2), "render_kernel_ms": round(tm["render_kernel_ms"], 2), "ms_per_window": round(wall / K, 4),
       "grays_s": round(c["rays"] / wall / 1e6, 2), "rays_per_window": c["rays"] // K}
if has:
    L.crt_debug_pool_density(buf, 0)
    d = [int(v) for v in buf]
    per = lambda a, b: round(d[a] / d[b], 2) if d[b] else None
    out["density"] = {
        "trips_per_window_M": round(d[0] / K / 1e6, 3), "resident_lanes_per_trip": per(22, 0),
        "node_runs_per_trip": per(1, 0), "node_lanes_per_run": per(2, 1), "tri_runs_per_trip": per(3, 0), "tri_lanes_per_run": per(4, 3),
        "tlas_runs_per_trip": per(23, 0), "tlas_lanes_per_run": per(24, 23),
        "swap_out_runs_per_trip": per(5, 0), "swap_out_lanes_per_run": per(6, 5), "swap_in_runs_per_trip": per(7, 0), "swap_in_lanes_per_run": per(8, 7), "load_issues_per_trip": per(21, 0),
        "end_passes_per_window_M": round(d[9] / K / 1e6, 3), "end_lanes_per_pass": per(10, 9), "end_miss_lanes_per_pass": per(11, 9), "end_gen_lanes_per_pass": per(12, 9),
        "end_depth_gt0_per_pass": per(25, 9), "end_depth_gt1_per_pass": per(26, 9), "end_depth_gt2_per_pass": per(27, 9),
        "bounce_passes_per_window_M": round(d[14] / K / 1e6, 3), "bounce_lanes_per_pass": per(15, 14), "bounce_mesh_lanes_per_pass": per(16, 14),
        "rejection_wave_iterations_per_pass": per(17, 14), "rejection_lanes_per_iteration": per(18, 17),
        "new_ray_lanes_M_per_window": round(d[20] / K / 1e6, 3), "new_ray_walk_fraction": per(19, 20), "raw": d}
print(json.dumps(out))
