#!/usr/bin/env python3
"""Workload for profiling the kernels bench.py does not exercise (run under rocprofv3 by tools/profile_other_kernels.sh):
whitted_kernel (config 1: cube, 640x360), find_nearest_kernel (1 M rays, bunny and TLAS scene), find_nearest_kd_kernel / find_nearest_grid_kernel."""
import importlib.util, os, sys, time
os.environ.setdefault("CRT_ENABLE_DEBUG_HOOKS", "1")      # the library reads its diagnostic environment switches only for processes that opt in
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("cpu_ray_tracer_amd", os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
crt = importlib.util.module_from_spec(spec); spec.loader.exec_module(crt)
A = os.path.join(REPO, "assets")
rng = np.random.default_rng(7)
n = 1 << 20
O = rng.uniform(-3, 3, (n, 3)).astype(np.float32); O[:, 1] = np.abs(O[:, 1]) + 0.2
D = (np.array([0, -0.3, 2], np.float32) + rng.uniform(-0.8, 0.8, (n, 3)).astype(np.float32)) - O
D = (D / np.linalg.norm(D, axis=1, keepdims=True)).astype(np.float32)
out = {}
cube = crt.HostScene(os.path.join(A, "scenes", "cube_scene.xml"), 0, A)
ctx = crt.Context(640, 360); cube.upload(ctx)
for i in range(5):
    t0 = time.perf_counter(); ctx.whitted_tick(); out["whitted_cube_640x360_ms"] = (time.perf_counter() - t0) * 1e3
ctx.close()
for xml, kind in (("bunny_scene.xml", 0), ("tlas_scene.xml", 1)):
    hs = crt.HostScene(os.path.join(A, "scenes", xml), kind, A)
    ctx = crt.Context(64, 64); hs.upload(ctx)
    for i in range(3):
        t0 = time.perf_counter(); h = ctx.find_nearest(O, D); out["find_nearest_%s_1Mrays_ms_incl_copies" % xml] = (time.perf_counter() - t0) * 1e3
    if kind == 0:
        for code, name in ((crt.ACCEL_KDTREE, "kd"), (crt.ACCEL_GRID, "grid")):
            hs.build_alt(code); hs.upload_alt(ctx, code)
            for i in range(3):
                t0 = time.perf_counter(); ctx.find_nearest_alt(code, O, D); out["find_nearest_%s_bunny_1Mrays_ms_incl_copies" % name] = (time.perf_counter() - t0) * 1e3
    ctx.close()
print(out)
