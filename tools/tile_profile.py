#!/usr/bin/env python3
"""Per-tile wall-time map of render_tiles_kernel (collectStats context): shows load imbalance between tiles and the
dispatch timeline.  Usage: python tools/tile_profile.py [scene.xml kind W H spp]"""
import importlib.util, os, sys
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
spp = int(sys.argv[5]) if len(sys.argv) > 5 else 64
sc = crt.HostScene(os.path.join(A, "scenes", xml), kind, A)
ctx = crt.Context(W, H, collect_stats=True)
sc.upload(ctx)
ctx.render(1, spp, 1); ctx.sync()
ctx.render(1, spp, 1); ctx.sync()
tm = ctx.timing()
n = (W // 16) * (H // 16)
tc = ctx.tile_clocks(n).astype(np.float64)
dur = tc[:, 0] / 100.0   # us (100 MHz)
trips = tc[:, 1]
print("kernel ms", tm["render_kernel_ms"])
print("tile wall us: min %.0f  median %.0f  mean %.0f  p90 %.0f  p99 %.0f  max %.0f" % (dur.min(), np.median(dur), dur.mean(), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
print("sum of tile wall time / kernel time = %.1f concurrent waves on average" % (dur.sum() / (tm["render_kernel_ms"] * 1e3)))
print("loop trips per tile: median %.0f mean %.0f max %.0f ; us per trip (heaviest tiles) %.3f" % (np.median(trips), trips.mean(), trips.max(), (dur[np.argsort(-dur)[:16]] / np.maximum(trips[np.argsort(-dur)[:16]], 1)).mean()))
c = ctx.counters()
print("counters (2 launches):", c)
steps = (c["interior_iters"] + c["tri_tests"] + c["tlas_iters"] + c["rays"]) / 2
print("lane-steps per launch %.3g ; wave trips per launch %.3g ; lanes advanced per trip %.1f" % (steps, trips.sum(), steps / trips.sum()))
tw = W // 16
m = dur.reshape(H // 16, tw)
print("row means (us):", " ".join("%.0f" % v for v in m.mean(axis=1)))
late = np.argsort(-dur)[:6]
for t in late:
    print("  tile %d (tx %d ty %d): dur %.0f us trips %.0f" % (t, t % tw, t // tw, dur[t], trips[t]))
