#!/usr/bin/env python3
"""Does splitting a job into several concurrent launches (crt_config.maxFramesPerLaunch) let the ordered accumulate of the early launches overlap the later ones?
(No: equal-priority launches share the machine and end together — 2 x 32 windows 221.2 ms like 1 x 64, smaller pieces are slower.)  Bit-checked by CRC."""
import importlib.util, os, sys, time, zlib
os.environ.setdefault("CRT_ENABLE_DEBUG_HOOKS", "1")      # the library reads its diagnostic environment switches only for processes that opt in
import numpy as np
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
spec = importlib.util.spec_from_file_location("cpu_ray_tracer_amd", os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
crt = importlib.util.module_from_spec(spec); spec.loader.exec_module(crt)
A = os.path.join(REPO, "assets")
sc = crt.HostScene(os.path.join(A, "scenes", "bunny_scene.xml"), 0, A)
for K in (64, 20):
    for mfl in (4096, 2048, 1024, 3072, 512):
        if mfl >= 64 * K and mfl != 4096: continue
        ctx = crt.Context(1280, 720, max_frames_per_launch=mfl); sc.upload(ctx); ctx.reserve(64 * K, 1)
        ts = []
        for i in range(4):
            ctx.clear(); ctx.sync(); t0 = time.perf_counter(); ctx.render(1, 64 * K, 1); ctx.sync(); ts.append((time.perf_counter() - t0) * 1e3)
        tm = ctx.timing()
        print("K=%d max frames per launch %4d: %.1f ms (runs %s) launches %d crc %08x" % (K, mfl, min(ts[1:]), " ".join("%.1f" % t for t in ts), tm["render_launches"], zlib.crc32(ctx.accumulator().tobytes())), flush=True)
        ctx.close()
