#!/usr/bin/env python3
"""Diagnostic: how good is the cost probe of the latency mode?  Dumps, for one camera, the probe's per-tile estimate, the one-wavefront-per-tile cost measured without the probe,
and the settled stage's lanes per tile, to a .npz; prints their relation.
    python tools/probe_quality.py out.npz [scene.xml kind W H]"""
import ctypes as C, importlib.util, json, os, sys, time
os.environ.setdefault("CRT_ENABLE_DEBUG_HOOKS", "1")
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("cpu_ray_tracer_amd", os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
crt = importlib.util.module_from_spec(spec); spec.loader.exec_module(crt)
A = os.path.join(REPO, "assets")
out = sys.argv[1]; a = sys.argv[2:]
xml, kind = (a[0], int(a[1])) if len(a) > 1 else ("bunny_scene.xml", 0)
W, H = (int(a[2]), int(a[3])) if len(a) > 3 else (1280, 720)
sc = crt.HostScene(os.path.join(A, "scenes", xml), kind, A)
L = crt.lib()
def stage(ctx, k):
    n = ctx.tile_count if hasattr(ctx, "tile_count") else (W // 16) * (H // 16)
    lanes = np.zeros(n, np.uint8); cost = np.zeros(n, np.uint32)
    r = L.crt_debug_lat_stage(ctx.h, k, lanes.ctypes.data_as(C.c_void_p), cost.ctypes.data_as(C.c_void_p))
    return (r, lanes, cost)
def render(ctx):
    ctx.clear(); ctx.sync(); t0 = time.perf_counter(); ctx.render(1, 64, 1); ctx.sync(); return (time.perf_counter() - t0) * 1e3
res = {}
# (a) no probe: stage 0 = one wavefront per tile, measured
os.environ["CRT_LAT_NO_PROBE"] = "1"
ctx = crt.Context(W, H); sc.upload(ctx); ctx.reserve(64, 1)
ts = [render(ctx) for _ in range(3)]
r, l0, c0 = stage(ctx, 0)
res["noprobe_ms"] = [round(t, 2) for t in ts]
ctx.close(); del os.environ["CRT_LAT_NO_PROBE"]
# (b) default: probe, then the tuner's stages
ctx = crt.Context(W, H); sc.upload(ctx); ctx.reserve(64, 1)
ts = [render(ctx)]
r, lp, est = stage(ctx, 0)
ts += [render(ctx) for _ in range(19)]
res["probed_ms"] = [round(t, 2) for t in ts]
best = None
for k in range(6, -1, -1):
    r, lk, ck = stage(ctx, k)
    if r >= 0 and r == k: best = (k, lk, ck); break
np.savez(out, onewave=c0, est=est, probe_lanes=lp, settled_stage=best[0] if best else -1, settled_lanes=best[1] if best else l0, settled_cost=best[2] if best else c0)
m = c0 > 0
ratio = est[m] / c0[m]; heavy = c0 >= 0.3 * c0.max()
res.update({"tiles": int(len(c0)), "onewave_top_ms": round(c0.max() * 1e-5, 2), "heavy_tiles(>=0.3 top)": int(heavy.sum()),
            "est/onewave median": float(np.median(ratio[heavy[m]])), "est/onewave cv among heavy": float(np.std(ratio[heavy[m]]) / np.mean(ratio[heavy[m]])),
            "settled_stage": int(best[0]) if best else -1, "settled_wavefronts": int((64 // best[1].astype(int)).sum()) if best else None,
            "settled_lanes_hist": {str(k): int((best[1] == k).sum()) for k in (64, 32, 16, 8, 4, 2, 1)} if best else None,
            "probe_lanes_hist": {str(k): int((lp == k).sum()) for k in (64, 32, 16, 8, 4, 2, 1)}})
print(json.dumps(res))
