#!/usr/bin/env python3
"""Diagnostic (-DCRT_STAMPS build only): where the heaviest tiles' wavefronts spend their shader cycles per loop trip."""
import ctypes as C, importlib.util, os, sys
os.environ.setdefault("CRT_ENABLE_DEBUG_HOOKS", "1")      # the library reads its diagnostic environment switches only for processes that opt in
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("cpu_ray_tracer_amd", os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
crt = importlib.util.module_from_spec(spec); spec.loader.exec_module(crt)
A = os.path.join(REPO, "assets")
W, H = 1280, 720
sc = crt.HostScene(os.path.join(A, "scenes", "bunny_scene.xml"), 0, A)
ctx = crt.Context(W, H, collect_stats=True); sc.upload(ctx)
ctx.render(1, 64, 1); ctx.sync(); ctx.render(1, 64, 1); ctx.sync()
n = (W // 16) * (H // 16)
tc = ctx.tile_clocks(n).astype(np.float64)
st = np.zeros((n, 16), np.uint64)
assert ctx.L.crt_debug_tile_stamps(ctx.h, st.ctypes.data_as(C.c_void_p)) == 0
st = st.astype(np.float64)
heavy = np.argsort(-tc[:, 0])[:32]
names = ["wait(vmcnt)", "shade", "tlas+node", "tri", "load issue", "total"]
trips = tc[heavy, 1].mean()
print("32 heaviest tiles: mean wall %.0f us, trips %.0f" % (tc[heavy, 0].mean() / 100, trips))
for i, nm in enumerate(names):
    print("  %-12s %10.0f cycles/trip" % (nm, st[heavy, i].mean() / trips))
print("  runs per trip: shade %.3f node %.3f tri %.3f" % tuple(st[heavy, 6 + i].mean() / trips for i in range(3)))
print("  cycles per shade run %.0f, per tri run %.0f" % (st[heavy, 1].mean() / st[heavy, 6].mean(), st[heavy, 3].mean() / st[heavy, 8].mean()))

# whole image (throughput view): where the issue slots of ALL waves go
tot = st.sum(axis=0)
print("all %d tiles: trips %.0f; cycles share: wait %.1f%% shade %.1f%% node %.1f%% tri %.1f%% load-issue %.1f%%" % ((n, tot[9]) + tuple(100 * tot[i] / tot[5] for i in range(5))))
for i, nm in enumerate(["shade", "node", "tri"]):
    runs, lanes = tot[6 + i], tot[10 + i]
    print("  %-5s runs/trip %.3f  mean lanes/run %.1f (%.0f%% of 64)  cycles/run %.0f" % (nm, runs / tot[9], lanes / max(runs, 1), 100 * lanes / max(runs, 1) / 64, tot[[1, 2, 3][i]] / max(runs, 1)))
