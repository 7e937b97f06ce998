#!/usr/bin/env python3
"""Diagnostic (-DCRT_POOL_STAMPS build, CRT_LIB_PATH=...): share of render_pool_kernel's wave time per section of its loop, one pool-only job.
    python tools/pool_stamps.py [scene.xml kind [windows [W H]]]"""
import ctypes as C, importlib.util, json, os, sys
os.environ.setdefault("CRT_ENABLE_DEBUG_HOOKS", "1")      # the library reads its diagnostic environment switches only for processes that opt in
os.environ.setdefault("CRT_RENDER_KERNEL", "pool_always"); os.environ.setdefault("CRT_SPLIT_OFF", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("cpu_ray_tracer_amd", os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
crt = importlib.util.module_from_spec(spec); spec.loader.exec_module(crt)
A = os.path.join(REPO, "assets")
xml = sys.argv[1] if len(sys.argv) > 1 else "bunny_scene.xml"; kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
K = int(sys.argv[3]) if len(sys.argv) > 3 else 32
W = int(sys.argv[4]) if len(sys.argv) > 4 else 1280; H = int(sys.argv[5]) if len(sys.argv) > 5 else 720
sc = crt.HostScene(os.path.join(A, "scenes", xml), kind, A)
ctx = crt.Context(W, H); sc.upload(ctx); ctx.reserve(64 * K, 1)
ctx.render(1, 64 * K, 1); ctx.sync()
L = crt.lib(); buf = (C.c_uint64 * 16)()
L.crt_debug_pool_stamps(buf, 1)
ctx.clear(); ctx.render(1, 64 * K, 1); ctx.sync()
L.crt_debug_pool_stamps(buf, 0)
d = [int(v) for v in buf]
names = ["walk+swap_out", "swap_in+loads+decide", "END head (state, factors, sky, ray gen)", "END new_ray", "END tail (unwind, store)", "BOUNCE head (hit info, albedo)",
         "BOUNCE material draw + rejection loop", "BOUNCE normalise + factor store", "BOUNCE new_ray"]
tot = sum(d[:9])
print(json.dumps({"scene": xml, "windows": K, "trips_per_window_M": round(d[9] / K / 1e6, 3), "end_passes_per_window_M": round(d[10] / K / 1e6, 3), "bounce_passes_per_window_M": round(d[11] / K / 1e6, 3),
                  "share_pct": {n: round(100.0 * v / tot, 2) for n, v in zip(names, d[:9])},
                  "clocks_per_run": {"trip (walk+swap_out)": round(d[0] / d[9]), "trip (swap_in+loads+decide)": round(d[1] / d[9]), "END head": round(d[2] / max(d[10], 1)), "END new_ray": round(d[3] / max(d[10], 1)),
                                     "END tail": round(d[4] / max(d[10], 1)), "BOUNCE head": round(d[5] / max(d[11], 1)), "BOUNCE draw+rejection": round(d[6] / max(d[11], 1)),
                                     "BOUNCE normalise+store": round(d[7] / max(d[11], 1)), "BOUNCE new_ray": round(d[8] / max(d[11], 1))}}))
