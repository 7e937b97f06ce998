#!/usr/bin/env python3
"""Child of tools/dup_profile.sh: renders two 64-frame launches of the bench scene with the library named by CRT_LIB_PATH."""
import importlib.util, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("cpu_ray_tracer_amd", os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
crt = importlib.util.module_from_spec(spec); spec.loader.exec_module(crt)
A = os.path.join(REPO, "assets")
xml = sys.argv[1] if len(sys.argv) > 1 else "bunny_scene.xml"
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sc = crt.HostScene(os.path.join(A, "scenes", xml), kind, A)
ctx = crt.Context(1280, 720, render_streams=1); sc.upload(ctx)
for i in range(2):
    ctx.render(1, 64, 1); ctx.sync()
