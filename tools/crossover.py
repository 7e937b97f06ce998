#!/usr/bin/env python3
"""Diagnostic: job time of render_tiles_kernel vs render_pool_kernel for jobs of 2..64 windows (where should the back end switch?)."""
import importlib.util, os, sys, time
os.environ.setdefault("CRT_ENABLE_DEBUG_HOOKS", "1")      # the library reads its diagnostic environment switches only for processes that opt in
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("cpu_ray_tracer_amd", os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
crt = importlib.util.module_from_spec(spec); spec.loader.exec_module(crt)
A = os.path.join(REPO, "assets")
xml = sys.argv[1] if len(sys.argv) > 1 else "bunny_scene.xml"; kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1280; H = int(sys.argv[4]) if len(sys.argv) > 4 else 720
sc = crt.HostScene(os.path.join(A, "scenes", xml), kind, A)
for k in ("tiles", "pool_always"):
    os.environ["CRT_RENDER_KERNEL"] = k
    ctx = crt.Context(W, H); sc.upload(ctx); ctx.reserve(4096, 1)
    row = []
    for wnd in (2, 4, 8, 16, 32, 64):
        ts = []
        for i in range(2):
            ctx.clear(); ctx.sync(); t0 = time.perf_counter(); ctx.render(1, 64 * wnd, 1); ctx.sync(); ts.append((time.perf_counter() - t0) * 1e3)
        row.append("%d: %.1f" % (wnd, min(ts)))
    print(xml, k, " | ".join(row)); ctx.close()
