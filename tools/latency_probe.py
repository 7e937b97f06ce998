#!/usr/bin/env python3
"""Single-render latency (ONE WxH / 64-spp render: clear, crt_render(1, 64, 1), sync) for the narrow-wavefront settings of render_tiles_kernel's
latency mode (CRT_NARROW_LANES = 0 / 16 / 8 / 4 / 2), each checked bit for bit against the first.  Usage: python tools/latency_probe.py [scene.xml kind W H lanes,lanes,..]"""
import importlib.util, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("cpu_ray_tracer_amd", os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
crt = importlib.util.module_from_spec(spec); spec.loader.exec_module(crt)
A = os.path.join(REPO, "assets")
xml = sys.argv[1] if len(sys.argv) > 1 else "bunny_scene.xml"
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
H = int(sys.argv[4]) if len(sys.argv) > 4 else 720
sc = crt.HostScene(os.path.join(A, "scenes", xml), kind, A)
ref = None
for lanes in (sys.argv[5].split(",") if len(sys.argv) > 5 else ["0", "16", "8", "4", "2"]):
    os.environ["CRT_NARROW_LANES"] = lanes
    ctx = crt.Context(W, H); sc.upload(ctx)
    ts = []
    for i in range(5):
        ctx.clear(); ctx.sync(); t0 = time.perf_counter(); ctx.render(1, 64, 1); ctx.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    acc = ctx.accumulator(); ctx.close()
    if ref is None: ref = acc
    print("narrow lanes %2s: single 64-spp render %.2f ms (min %.2f)  identical: %s" % (lanes, sorted(ts)[2], min(ts), np.array_equal(acc, ref)))
