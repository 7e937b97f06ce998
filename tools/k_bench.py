#!/usr/bin/env python3
"""Diagnostic: pool-only job time at K = 20 and 64 windows for each library variant in build/variants (CRT_LIB_PATH), plus a CRC of the accumulator of a
4-window job (every variant must print the same one: scheduling never changes a pixel).
    python tools/k_bench.py [scene.xml kind [W H [K1 K2]]]        env: KB_ONLY=name1,name2 restricts the variants"""
import os, subprocess, sys
os.environ.setdefault("CRT_ENABLE_DEBUG_HOOKS", "1")      # the library reads its diagnostic environment switches only for processes that opt in
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V = os.path.join(REPO, "build", "variants")
child = r'''
import importlib.util, os, sys, time, zlib
REPO = %r
spec = importlib.util.spec_from_file_location("cpu_ray_tracer_amd", os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
crt = importlib.util.module_from_spec(spec); spec.loader.exec_module(crt)
A = os.path.join(REPO, "assets")
W, H = int(sys.argv[3]), int(sys.argv[4])
sc = crt.HostScene(os.path.join(A, "scenes", sys.argv[1]), int(sys.argv[2]), A)
ctx = crt.Context(W, H); sc.upload(ctx); ctx.reserve(64 * int(sys.argv[6]), 1)
ctx.render(1, 256, 1); ctx.sync(); crc = zlib.crc32(ctx.accumulator().tobytes())
out = ["crc %%08x" %% crc]
for K in (int(sys.argv[5]), int(sys.argv[6])):
    ts = []
    for i in range(3):
        ctx.clear(); ctx.sync(); t0 = time.perf_counter(); ctx.render(1, 64 * K, 1); ctx.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    out.append("K=%%d: %%.1f ms (%%.3f ms/step)" %% (K, min(ts), min(ts) / K))
print(" | ".join(out))
''' % REPO
a = sys.argv[1:]
scene, kind = (a[0], a[1]) if len(a) > 1 else ("bunny_scene.xml", "0")
W, H = (a[2], a[3]) if len(a) > 3 else ("1280", "720")
K1, K2 = (a[4], a[5]) if len(a) > 5 else ("20", "64")
only = [x for x in os.environ.get("KB_ONLY", "").split(",") if x]
for f in sorted(os.listdir(V)):
    if not f.endswith(".so") or (only and f[7:-3] not in only):
        continue
    env = dict(os.environ, CRT_LIB_PATH=os.path.join(V, f), CRT_RENDER_KERNEL="pool_always", CRT_SPLIT_OFF=os.environ.get("CRT_SPLIT_OFF", "1"))
    try:
        r = subprocess.run([sys.executable, "-c", child, scene, kind, W, H, K1, K2], env=env, capture_output=True, text=True, timeout=240)
        print("%-20s %s %s" % (f[7:-3], r.stdout.strip(), r.stderr.strip()[-300:] if r.returncode else ""), flush=True)
    except subprocess.TimeoutExpired:
        print("%-20s TIMEOUT (killed) — stopping" % f[7:-3], flush=True)
        sys.exit(3)
