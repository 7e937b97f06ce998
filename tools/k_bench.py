#!/usr/bin/env python3
"""Diagnostic: job time at K = 20 and 64 windows for each library variant in build/variants (CRT_LIB_PATH)."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V = os.path.join(REPO, "build", "variants")
child = r'''
import importlib.util, os, sys, time
REPO = %r
spec = importlib.util.spec_from_file_location("cpu_ray_tracer_amd", os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
crt = importlib.util.module_from_spec(spec); spec.loader.exec_module(crt)
A = os.path.join(REPO, "assets")
sc = crt.HostScene(os.path.join(A, "scenes", sys.argv[1]), int(sys.argv[2]), A)
ctx = crt.Context(1280, 720); sc.upload(ctx); ctx.reserve(4096, 1)
out = []
for K in (20, 64):
    ts = []
    for i in range(3):
        ctx.clear(); ctx.sync(); t0 = time.perf_counter(); ctx.render(1, 64 * K, 1); ctx.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    out.append("K=%%d: %%.1f ms (%%.3f ms/step)" %% (K, min(ts), min(ts) / K))
print(" | ".join(out))
''' % REPO
for f in sorted(os.listdir(V)):
    if f.endswith(".so"):
        r = subprocess.run([sys.executable, "-c", child] + (sys.argv[1:3] if len(sys.argv) > 2 else ["bunny_scene.xml", "0"]), env=dict(os.environ, CRT_LIB_PATH=os.path.join(V, f), CRT_RENDER_KERNEL="pool_always"), capture_output=True, text=True)
        print("%-20s %s %s" % (f, r.stdout.strip(), r.stderr.strip()[-200:] if r.returncode else ""))
