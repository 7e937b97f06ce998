#!/usr/bin/env python3
"""Single-render latency (clear, crt_render(1, 64, 1), sync) over the tuner's whole sequence: first render, untuned, every stage, settled median.
    python tools/single_render.py [scene.xml kind W H [renders]]        env SR_SPLIT="r/n": only the tiles rank r of an n-way tile split owns"""
import importlib.util, json, os, sys, time
os.environ.setdefault("CRT_ENABLE_DEBUG_HOOKS", "1")      # the library reads its diagnostic environment switches only for processes that opt in
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("cpu_ray_tracer_amd", os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
crt = importlib.util.module_from_spec(spec); spec.loader.exec_module(crt)
A = os.path.join(REPO, "assets")
a = sys.argv[1:]
xml, kind = (a[0], int(a[1])) if len(a) > 1 else ("bunny_scene.xml", 0)
W, H = (int(a[2]), int(a[3])) if len(a) > 3 else (1280, 720)
n = int(a[4]) if len(a) > 4 else 20
sc = crt.HostScene(os.path.join(A, "scenes", xml), kind, A)
kw = {}
if os.environ.get("SR_SPLIT"):
    r_, n_ = (int(v) for v in os.environ["SR_SPLIT"].split("/")); first, stride, count = crt.tile_partition(r_, n_, (W // 16) * (H // 16)); kw = dict(tile_first=first, tile_stride=stride, tile_count=count)
ctx = crt.Context(W, H, **kw); sc.upload(ctx); ctx.reserve(64, 1)
# warm the process (code objects, slab pool) with another camera, then measure from a camera change on: first_ms = the first render after it
ctx.set_camera_state((0.3, 0.2, -2.2), (0.0, 0.0, 1.0)); ctx.render(1, 64, 1); ctx.sync()
ctx.set_camera_state((0.0, 0.0, -2.0), (0.0, 0.0, -1.0))
ts = []; ks = []; subs = []
k0 = ctx.timing()["render_kernel_ms"]
for i in range(n):
    ctx.clear(); ctx.sync(); t0 = time.perf_counter(); ctx.render(1, 64, 1); t1 = time.perf_counter(); ctx.sync(); ts.append((time.perf_counter() - t0) * 1e3); subs.append((t1 - t0) * 1e3)
    k1 = ctx.timing()["render_kernel_ms"]; ks.append(k1 - k0); k0 = k1
print(json.dumps({"scene": xml, "size": [W, H], "renders_ms": [round(t, 2) for t in ts], "render_kernel_ms": [round(t, 2) for t in ks], "submit_ms": [round(t, 2) for t in subs], "first_ms": round(ts[0], 2), "settled_ms": round(float(np.median(ts[-5:])), 2), "best_ms": round(min(ts), 2)}))
