#!/usr/bin/env python3
"""Job time (one crt_render of K windows, 64 spp each) with and without split launches (CRT_SPLIT_OFF / CRT_SPLIT_SLACK) and with either kernel forced, bit-checked
against each other.  Usage: python tools/split_probe.py [K,K,..] [scene.xml kind W H]"""
import importlib.util, os, subprocess, sys, time
os.environ.setdefault("CRT_ENABLE_DEBUG_HOOKS", "1")      # the library reads its diagnostic environment switches only for processes that opt in
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import importlib.util, os, sys, time, zlib
import numpy as np
REPO = %r
spec = importlib.util.spec_from_file_location("cpu_ray_tracer_amd", os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
crt = importlib.util.module_from_spec(spec); spec.loader.exec_module(crt)
A = os.path.join(REPO, "assets")
K, xml, kind, W, H = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
sc = crt.HostScene(os.path.join(A, "scenes", xml), kind, A)
stride = int(os.environ.get("PROBE_TILE_STRIDE", "1")); tiles = (W // 16) * (H // 16)          # (stride N: the tiles one rank of an N-GPU tile split owns)
ctx = crt.Context(W, H, tile_stride=stride, tile_count=(tiles + stride - 1) // stride if stride > 1 else -1); sc.upload(ctx); ctx.reserve(64 * K, 1)
ts = []
for i in range(4):
    ctx.clear(); ctx.sync(); t0 = time.perf_counter(); ctx.render(1, 64 * K, 1); ctx.sync(); ts.append((time.perf_counter() - t0) * 1e3); tm = ctx.timing()
print("%%7.1f ms (%%.3f ms/step; runs %%s; last: %%d launches, %%d pool, %%d split)  crc %%08x" %% (min(ts[1:]), min(ts[1:]) / K, " ".join("%%.1f" %% t for t in ts), tm["render_launches"], tm["pool_launches"], tm["split_launches"], zlib.crc32(ctx.accumulator().tobytes())))
''' % REPO
Ks = [int(k) for k in (sys.argv[1] if len(sys.argv) > 1 else "20").split(",")]
scene = sys.argv[2:6] if len(sys.argv) > 5 else ["bunny_scene.xml", "0", "1280", "720"]
for K in Ks:
    variants = [("default", {}), ("no split", {"CRT_SPLIT_OFF": "1"}), ("tiles kernel", {"CRT_RENDER_KERNEL": "tiles"})]
    for sl in os.environ.get("PROBE_NARROW_SLOTS", "").split(","):
        if sl: variants.append(("narrow slots " + sl, {"CRT_PLAN_NARROW_SLOTS": sl}))
    for name, env in variants:
        r = subprocess.run([sys.executable, "-c", child, str(K)] + scene, env=dict(os.environ, **env), capture_output=True, text=True)
        print("K = %2d  %-26s %s %s" % (K, name, r.stdout.strip(), r.stderr.strip()[-300:] if r.returncode else ""), flush=True)
        crt_lines = [l for l in r.stderr.splitlines() if l.startswith("[crt] job")]
        if crt_lines: print("        " + crt_lines[-1])
