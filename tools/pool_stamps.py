#!/usr/bin/env python3
"""Diagnostic (-DCRT_POOL_STAMPS build, CRT_LIB_PATH=...): share of render_pool_kernel's wave time per section of its loop, one 16-window job."""
import importlib.util, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("cpu_ray_tracer_amd", os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
crt = importlib.util.module_from_spec(spec); spec.loader.exec_module(crt)
A = os.path.join(REPO, "assets")
xml = sys.argv[1] if len(sys.argv) > 1 else "bunny_scene.xml"; kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sc = crt.HostScene(os.path.join(A, "scenes", xml), kind, A)
ctx = crt.Context(1280, 720); sc.upload(ctx)
ctx.render(1, 1024, 1); ctx.sync(); ctx.reset_counters(); ctx.render(1, 1024, 1); ctx.sync()
c = ctx.counters(); tm = ctx.timing()
walk, swap, end, bnc, trips = c["interior_iters"], c["leaf_iters"], c["tri_tests"], c["tlas_iters"], c["blas_visits"]
tot = walk + swap + end + bnc
print("%s: trips %.2fM per window; wave time: walk + swap + loads %.1f%% END passes %.1f%% BOUNCE passes %.1f%%; cycles/trip: walk+swap %.0f end %.0f bounce %.0f"
      % (xml, trips / 16e6, 100 * (walk + swap) / tot, 100 * end / tot, 100 * bnc / tot, (walk + swap) / trips, end / trips, bnc / trips))
