#!/usr/bin/env python3
"""Diagnostic (-DCRT_POOL_TIMELINE build, CRT_LIB_PATH=...): wavefronts of render_pool_kernel in flight over time for a K-window job, and what the job's
tail looks like.  Usage: python tools/pool_timeline.py [K] [scene.xml kind]"""
import ctypes as C, importlib.util, os, sys
os.environ.setdefault("CRT_ENABLE_DEBUG_HOOKS", "1")      # the library reads its diagnostic environment switches only for processes that opt in
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("cpu_ray_tracer_amd", os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
crt = importlib.util.module_from_spec(spec); spec.loader.exec_module(crt)
A = os.path.join(REPO, "assets")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
xml = sys.argv[2] if len(sys.argv) > 2 else "bunny_scene.xml"; kind = int(sys.argv[3]) if len(sys.argv) > 3 else 0
os.environ.setdefault("CRT_RENDER_KERNEL", "pool_always")
sc = crt.HostScene(os.path.join(A, "scenes", xml), kind, A)
ctx = crt.Context(1280, 720); sc.upload(ctx)
ctx.render(1, 64 * K, 1); ctx.sync(); ctx.clear(); ctx.render(1, 64 * K, 1); ctx.sync()
tm = ctx.timing()
L = ctx.L; L.crt_debug_pool_timeline.restype = C.c_size_t; L.crt_debug_pool_timeline.argtypes = [C.c_void_p, C.c_size_t]
n = L.crt_debug_pool_timeline(None, 0)
buf = np.zeros((n, 3), np.uint64); L.crt_debug_pool_timeline(buf.ctypes.data_as(C.c_void_p), n)
t0 = buf[:, 0].astype(np.float64); t1 = buf[:, 1].astype(np.float64)
base = t0.min(); t0 = (t0 - base) * 1e-5; t1 = (t1 - base) * 1e-5          # ms
dur = t1 - t0
print("K = %d: %d wavefronts, launch %.1f ms by HIP events, last wavefront ends at %.1f ms" % (K, n, tm["render_kernel_ms"] / max(tm["render_launches"], 1), t1.max()))
print("wavefront duration: mean %.2f ms, median %.2f, p90 %.2f, p99 %.2f, max %.2f; sum %.0f wave-ms = %.1f ms x 4096 slots" % (dur.mean(), np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max(), dur.sum(), dur.sum() / 4096))
edges = np.linspace(0, t1.max(), 41)
for a, b in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a + b)
    live = int(((t0 <= mid) & (t1 > mid)).sum())
    started = int(((t0 >= a) & (t0 < b)).sum())
    print("  %6.1f ms: %5d in flight %s  (+%d started)" % (mid, live, "#" * (live // 100), started))
order = np.argsort(-t1)[:8]
print("last to finish (block, start, duration): " + ", ".join("%d: %.1f + %.1f" % (i, t0[i], dur[i]) for i in order))
lo = np.argsort(-dur)[:8]
print("longest (block, start, duration): " + ", ".join("%d: %.1f + %.1f" % (i, t0[i], dur[i]) for i in lo))
