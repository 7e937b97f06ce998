#!/usr/bin/env python3
"""Diagnostic (-DCRT_DUO_STATS build of the whole library, CRT_LIB_PATH=...): how busy the two wavefronts of render_duo_kernel are.   python tools/duo_stats.py [K [scene kind W H]]"""
import ctypes as C, importlib.util, json, os, sys, time
os.environ.setdefault("CRT_ENABLE_DEBUG_HOOKS", "1"); os.environ.setdefault("CRT_RENDER_KERNEL", "pool_always"); os.environ.setdefault("CRT_SPLIT_OFF", "1"); os.environ.setdefault("CRT_POOL_DUO", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("cpu_ray_tracer_amd", os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
crt = importlib.util.module_from_spec(spec); spec.loader.exec_module(crt)
A = os.path.join(REPO, "assets")
a = sys.argv[1:]; K = int(a[0]) if a else 32
xml, kind, W, H = (a[1], int(a[2]), int(a[3]), int(a[4])) if len(a) > 4 else ("bunny_scene.xml", 0, 1280, 720)
sc = crt.HostScene(os.path.join(A, "scenes", xml), kind, A)
ctx = crt.Context(W, H); sc.upload(ctx); ctx.reserve(64 * K, 1)
ctx.render(1, 64 * K, 1); ctx.sync()
L = crt.lib(); buf = (C.c_uint64 * 16)(); L.crt_debug_duo_stats(buf, 1)
ctx.clear(); ctx.sync(); t0 = time.perf_counter(); ctx.render(1, 64 * K, 1); ctx.sync(); ms = (time.perf_counter() - t0) * 1e3
L.crt_debug_duo_stats(buf, 0); d = [int(v) for v in buf]
per = lambda x, y: round(d[x] / d[y], 2) if d[y] else None
print(json.dumps({"K": K, "job_ms": round(ms, 1), "trips_per_window_M": round(d[0] / K / 1e6, 3), "lanes_per_trip": per(1, 0), "second_steps_per_trip": per(12, 0), "lanes_per_second_step": per(13, 12),
                  "walker_idle_polls_per_window_M": round(d[2] / K / 1e6, 3), "end_passes_per_window_M": round(d[3] / K / 1e6, 3), "end_lanes": per(4, 3), "bounce_passes_per_window_M": round(d[5] / K / 1e6, 3), "bounce_lanes": per(6, 5),
                  "shader_idle_polls_per_window_M": round(d[7] / K / 1e6, 3), "walker_busy_frac": round(d[8] / max(d[8] + d[9], 1), 3), "shader_busy_frac": round(d[10] / max(d[10] + d[11], 1), 3)}))
