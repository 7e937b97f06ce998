import importlib.util, os, sys, time, numpy as np
os.environ.setdefault("CRT_ENABLE_DEBUG_HOOKS", "1")      # the library reads its diagnostic environment switches only for processes that opt in
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
spec = importlib.util.spec_from_file_location("cpu_ray_tracer_amd", os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
crt = importlib.util.module_from_spec(spec); spec.loader.exec_module(crt)
A = os.path.join(REPO, "assets")
for xml, kind in [("bunny_scene.xml", 0), ("tlas_scene.xml", 1)]:
    sc = crt.HostScene(os.path.join(A, "scenes", xml), kind, A)
    ctx = crt.Context(1280, 720); sc.upload(ctx)
    for frames in (1, 4, 16, 64):
        ts = []
        for i in range(6):
            t0 = time.perf_counter(); ctx.render(1 + i * frames, frames, 1); ctx.sync(); ts.append((time.perf_counter() - t0) * 1e3)
        print(xml, "frames per call", frames, "latency ms %.2f" % np.median(ts[1:]))
    r = crt.HostRenderer(sc, 1280, 720); r.init()
    ts = []
    for i in range(6):
        t0 = time.perf_counter(); r.tick(0.0); ts.append((time.perf_counter() - t0) * 1e3)
    print(xml, "Renderer::Tick (1 spp + accumulator/screen read-back) ms %.2f" % np.median(ts[1:]))
