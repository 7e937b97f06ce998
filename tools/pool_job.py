#!/usr/bin/env python3
"""One pool-only job of K windows (default 32) of a scene, twice (warm-up + timed); prints a JSON line with the timed job's duration and ray counter.
The program the PMC passes of tools/variant_pmc.sh profile.   python3 tools/pool_job.py [K scene.xml kind W H]"""
import importlib.util, json, os, sys, time
os.environ.setdefault("CRT_ENABLE_DEBUG_HOOKS", "1")      # the library reads its diagnostic environment switches only for processes that opt in
os.environ.setdefault("CRT_RENDER_KERNEL", "pool_always"); os.environ.setdefault("CRT_SPLIT_OFF", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("cpu_ray_tracer_amd", os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
crt = importlib.util.module_from_spec(spec); spec.loader.exec_module(crt)
A = os.path.join(REPO, "assets")
a = sys.argv[1:]
K = int(a[0]) if a else 32
xml, kind = (a[1], int(a[2])) if len(a) > 2 else ("bunny_scene.xml", 0)
W, H = (int(a[3]), int(a[4])) if len(a) > 4 else (1280, 720)
sc = crt.HostScene(os.path.join(A, "scenes", xml), kind, A)
ctx = crt.Context(W, H); sc.upload(ctx); ctx.reserve(64 * K, 1)
ctx.render(1, 64 * K, 1); ctx.sync(); ctx.timing(); ctx.reset_counters(); ctx.clear(); ctx.sync()
t0 = time.perf_counter(); ctx.render(1, 64 * K, 1); ctx.sync(); wall = (time.perf_counter() - t0) * 1e3
tm = ctx.timing(); c = ctx.counters()
print(json.dumps({"windows": K, "scene": xml, "job_ms": round(wall, 3), "render_kernel_ms": round(tm["render_kernel_ms"], 3), "rays_per_window": c["rays"] // K,
                  "grays_s": round(c["rays"] / wall / 1e6, 3), "lib": os.path.basename(crt.LIB_PATH)}))
