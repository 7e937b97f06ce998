#!/usr/bin/env python3
"""Scheduler robustness: job times of the planner against the plain launches on cost distributions its constants were NOT fitted to.
For every (scene, size, camera) and job length K: planned (block table + pool split from measured tile costs, trial off), plain (cost-ordered, kernel by the
size rule, no table), all-pool, all-tiles, and auto (the shipped behaviour: planned first, plain second, the faster kept — measured from the fourth launch on).
Each variant in its own process; min of the repeats; all variants render the same pixels (CRC).   python tools/plan_probe.py > profiles/r03_plan_probe.json"""
import json, os, subprocess, sys
os.environ.setdefault("CRT_ENABLE_DEBUG_HOOKS", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import importlib.util, json, os, sys, time, zlib
REPO = %r
spec = importlib.util.spec_from_file_location("cpu_ray_tracer_amd", os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
crt = importlib.util.module_from_spec(spec); spec.loader.exec_module(crt)
A = os.path.join(REPO, "assets")
cfg = json.loads(sys.argv[1])
sc = crt.HostScene(os.path.join(A, "scenes", cfg["xml"]), cfg["kind"], A)
ctx = crt.Context(cfg["W"], cfg["H"]); sc.upload(ctx)
if cfg.get("cam"): ctx.set_camera_state(tuple(cfg["cam"][0]), tuple(cfg["cam"][1]))
K = cfg["K"]; ctx.reserve(64 * K, 1)
ts = []
for i in range(cfg["reps"]):
    ctx.clear(); ctx.sync(); t0 = time.perf_counter(); ctx.render(1, 64 * K, 1); ctx.sync(); ts.append((time.perf_counter() - t0) * 1e3)
tm = ctx.timing()
crc = zlib.crc32(ctx.accumulator().tobytes())
print(json.dumps({"ms": [round(t, 2) for t in ts], "crc": "%%08x" %% crc, "split_launches": tm.get("split_launches", 0), "pool_launches": tm.get("pool_launches", 0)}))
''' % REPO
CASES = [
    dict(name="bunny 1280x720, default camera (the fitted case)", xml="bunny_scene.xml", kind=0, W=1280, H=720),
    dict(name="bunny 1280x720, camera 0.8 in front of the mesh (most tiles expensive)", xml="bunny_scene.xml", kind=0, W=1280, H=720, cam=[[0.0, -0.5, 1.1], [0.0, -0.6, 2.0]]),
    dict(name="two-level scene 1280x720, camera above looking down", xml="tlas_scene.xml", kind=1, W=1280, H=720, cam=[[0.0, 3.2, -0.5], [0.0, 0.0, 2.0]]),
    dict(name="watch-tower 1920x1080, default camera", xml="tower_scene_jpg.xml", kind=0, W=1920, H=1080),
    dict(name="bunny 3840x2160, default camera", xml="bunny_scene.xml", kind=0, W=3840, H=2160),
]
VARIANTS = {"planned": {"CRT_PLAN_NO_TRIAL": "1"}, "plain": {"CRT_SPLIT_OFF": "1"}, "all_pool": {"CRT_SPLIT_OFF": "1", "CRT_RENDER_KERNEL": "pool_always"},
            "all_tiles": {"CRT_SPLIT_OFF": "1", "CRT_RENDER_KERNEL": "tiles"}, "auto": {}}
out = []
for case in CASES:
    for K in ((8, 20) if case["W"] < 3000 else (4, 8)):
        row = {"case": case["name"], "windows": K}
        crcs = set()
        for vn, env in VARIANTS.items():
            cfg = dict(case, K=K, reps=6 if vn == "auto" else 4)
            r = subprocess.run([sys.executable, "-c", child, json.dumps(cfg)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                row[vn] = "failed: " + r.stderr.strip()[-200:]; continue
            d = json.loads(r.stdout.strip().splitlines()[-1]); crcs.add(d["crc"])
            # the first launch of a process measures the tile costs (never planned): the planner's figure is the best of the later ones; auto: launches 4..6 (decided)
            row[vn] = min(d["ms"][3:]) if vn == "auto" else min(d["ms"][1:])
            if vn in ("planned", "auto"): row[vn + "_split_launches"] = d["split_launches"]
        best_plain = min(v for k, v in row.items() if k in ("plain", "all_pool", "all_tiles") and isinstance(v, float))
        row["identical_pixels"] = len(crcs) == 1
        if isinstance(row.get("planned"), float): row["planned_vs_best_plain"] = round(row["planned"] / best_plain, 3)
        if isinstance(row.get("auto"), float): row["auto_vs_best_plain"] = round(row["auto"] / best_plain, 3); row["auto_vs_plain"] = round(row["auto"] / row["plain"], 3)
        out.append(row); print(json.dumps(row), file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
