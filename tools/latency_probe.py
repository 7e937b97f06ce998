#!/usr/bin/env python3
"""Single-render latency (ONE WxH / 64-spp render: clear, crt_render(1, 64, 1), sync) of render_tiles_kernel's latency mode: one wavefront per tile
against block tables built from the measured tile costs with different policies (CRT_LAT_POLICY = "share:lanes,..": tiles whose cost is at least `share` of
the most expensive tile's run as 64 / lanes wavefronts of `lanes` lanes), each checked bit for bit against the first.
Usage: python tools/latency_probe.py [scene.xml kind W H "policy;policy;.."]   ("off" = one wavefront per tile, "default" = the built-in policy)"""
import ctypes as C, importlib.util, os, sys, time
os.environ.setdefault("CRT_ENABLE_DEBUG_HOOKS", "1")      # the library reads its diagnostic environment switches only for processes that opt in
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("cpu_ray_tracer_amd", os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
crt = importlib.util.module_from_spec(spec); spec.loader.exec_module(crt)
A = os.path.join(REPO, "assets")
xml = sys.argv[1] if len(sys.argv) > 1 else "bunny_scene.xml"
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
H = int(sys.argv[4]) if len(sys.argv) > 4 else 720
policies = sys.argv[5].split(";") if len(sys.argv) > 5 else ["off", "default", "0.85:2,0.65:4,0.45:8,0.30:16"]
sc = crt.HostScene(os.path.join(A, "scenes", xml), kind, A)
ref = None; watch = None
for pol in policies:
    for k in ("CRT_LAT_OFF", "CRT_LAT_POLICY", "CRT_LAT_AIM", "CRT_LAT_FORCE", "CRT_LAT_RECORD"): os.environ.pop(k, None)
    os.environ["CRT_LAT_RECORD"] = "1"
    if pol == "off": os.environ["CRT_LAT_POLICY"] = "2:64"; os.environ["CRT_LAT_FORCE"] = "1"      # a table of one wave per tile
    elif pol.startswith("aim="): os.environ["CRT_LAT_AIM"] = pol[4:]
    elif pol != "default": os.environ["CRT_LAT_POLICY"] = pol; os.environ["CRT_LAT_FORCE"] = "1"
    stride = int(os.environ.get("PROBE_TILE_STRIDE", "1")); tiles = (W // 16) * (H // 16)      # (stride > 1: how does the mode behave with a lighter load)
    ctx = crt.Context(W, H, tile_stride=stride, tile_count=(tiles + stride - 1) // stride if stride > 1 else -1); sc.upload(ctx)
    ts = []
    for i in range(24):
        ctx.clear(); ctx.sync(); t0 = time.perf_counter(); ctx.render(1, 64, 1); ctx.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    acc = ctx.accumulator()
    cost = np.zeros((W // 16) * (H // 16), np.uint32)       # (a strided context fills only its first tiles)
    top = ""
    if ctx.L.crt_debug_tile_costs(ctx.h, cost.ctypes.data_as(C.c_void_p)) == 0:
        o = np.argsort(-cost.astype(np.int64))[:3]
        top = "  slowest tiles " + ", ".join("%d: %.2f ms" % (t, cost[t] * 1e-5) for t in o)
        if watch is None: watch = o
        else: top += " | the one-wave launch's slowest now " + ", ".join("%d: %.2f" % (t, cost[t] * 1e-5) for t in watch)
    ctx.close()
    if ref is None: ref = acc
    print("%-34s renders %s ms; last four median %.2f  identical: %s" % (pol, " ".join("%.1f" % t for t in ts), sorted(ts[-4:])[1], np.array_equal(acc, ref)) + top, flush=True)
