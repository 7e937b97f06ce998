#!/usr/bin/env python3
"""Diagnostic (-DCRT_STAMPS build, CRT_LIB_PATH=...): dumps the per-tile clocks / phase stamps of one 64-frame window to an .npz."""
import ctypes as C, importlib.util, os, sys
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
out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(REPO, "gpurun_out", "stamps.npz")
sc = crt.HostScene(os.path.join(A, "scenes", xml), kind, A)
ctx = crt.Context(W, H, collect_stats=True); sc.upload(ctx)
ctx.render(1, 64, 1); ctx.sync(); ctx.render(1, 64, 1); ctx.sync()
n = (W // 16) * (H // 16)
tc = ctx.tile_clocks(n)
st = np.zeros((n, 16), np.uint64)
assert ctx.L.crt_debug_tile_stamps(ctx.h, st.ctypes.data_as(C.c_void_p)) == 0
np.savez_compressed(out, tc=tc, st=st, W=W, H=H)
print("saved", out)
