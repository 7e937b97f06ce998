#!/usr/bin/env python3
"""A/B timing of kernel build variants: compiles cpu-ray-tracer_amd with extra -D flags into build/variants/
(in the authoring container: `python tools/ab_bench.py build NAME=-DFOO=1 ...`), and times them on the GPU box
(`python tools/ab_bench.py run`), each variant in its own process, median of the render kernel's HIP-event time."""
import json, os, subprocess, sys
os.environ.setdefault("CRT_ENABLE_DEBUG_HOOKS", "1")      # the library reads its diagnostic environment switches only for processes that opt in
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(REPO, "build", "variants")
PKG = os.path.join(REPO, "cpu-ray-tracer_amd")
SRC = ["csrc/device/kernels.hip", "csrc/device/render_pool.hip", "csrc/device/render_narrow.hip", "csrc/device/render_prim.hip", "csrc/device/alt_accel.hip", "csrc/abi.cpp", "csrc/host/accel.cpp", "csrc/host/accel_alt.cpp", "csrc/host/loaders.cpp", "csrc/host/scene.cpp", "csrc/host/primitive_scene.cpp", "csrc/host/host_abi.cpp"]

def fastbuild(specs):
    """variants that only differ in render_pool.hip: everything else is compiled once into build/obj/ and re-linked (seconds per variant)"""
    odir = os.path.join(REPO, "build", "obj"); os.makedirs(odir, exist_ok=True); os.makedirs(VDIR, exist_ok=True)
    base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Xarch_device", "-fno-slp-vectorize"]
    objs = []
    for src in SRC:
        if src.endswith("render_pool.hip"): continue
        o = os.path.join(odir, os.path.basename(src) + ".o"); objs.append(o)
        deps = [os.path.join(PKG, src)] + [os.path.join(PKG, "csrc/device", h) for h in ("dev_common.h", "layout.h")] + [os.path.join(REPO, "include", h) for h in ("crt_abi.h", "crt_host.h")]
        if not os.path.exists(o) or any(os.path.getmtime(d) > os.path.getmtime(o) for d in deps):
            subprocess.check_call(base + ["-c", "-o", o, src], cwd=PKG)
    procs = []
    for spec in specs:
        name, _, flags = spec.partition("=")
        po = os.path.join(odir, "pool_%s.o" % name)
        out = os.path.join(VDIR, "libcrt_%s.so" % name)
        src = "csrc/device/render_pool.hip"
        fl = flags.split()
        if fl and fl[0].startswith("@"):      # NAME=@/path/to/another/render_pool.hip [flags]: an older version of the kernel as a variant
            src = fl[0][1:]; fl = fl[1:] + ["-I" + os.path.join(PKG, "csrc/device")]
        cmd = " ".join(base + fl + ["-c", "-o", po, src]) + " && " + " ".join(base + ["-shared", "-o", out, po] + objs + ["-lz"])
        procs.append((name, subprocess.Popen(cmd, shell=True, cwd=PKG)))
        if len(procs) >= 6:
            for n, pr in procs:
                if pr.wait() != 0: sys.exit("variant %s failed to build" % n)
            procs = []
    for n, pr in procs:
        if pr.wait() != 0: sys.exit("variant %s failed to build" % n)

def build(specs):
    os.makedirs(VDIR, exist_ok=True)
    for spec in specs:
        name, _, flags = spec.partition("=")
        out = os.path.join(VDIR, "libcrt_%s.so" % name)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Xarch_device", "-fno-slp-vectorize", "-shared", "-o", out] + flags.split() + SRC + ["-lz"]
        print(" ".join(cmd)); subprocess.check_call(cmd, cwd=PKG)

def run(args):
    child = r'''
import importlib.util, os, sys, numpy as np
REPO = %r
spec = importlib.util.spec_from_file_location("cpu_ray_tracer_amd", os.path.join(REPO, "cpu-ray-tracer_amd", "__init__.py"))
crt = importlib.util.module_from_spec(spec); spec.loader.exec_module(crt)
A = os.path.join(REPO, "assets")
xml, kind, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sc = crt.HostScene(os.path.join(A, "scenes", xml), kind, A)
import time
ctx = crt.Context(W, H, render_streams=int(os.environ.get("AB_STREAMS", "7"))); sc.upload(ctx)
ts = []
for i in range(4):
    ctx.clear(); ctx.render(1, 64, 1); ctx.sync(); ts.append(ctx.timing()["render_kernel_ms"])
ctx.clear(); ctx.sync(); t0 = time.perf_counter()
for i in range(56): ctx.render(1 + 64 * i, 64, 1)
ctx.sync(); thr = (time.perf_counter() - t0) / 56 * 1e3
ctx.clear(); ctx.render(1, 64 * 64, 1); ctx.sync(); ctx.timing()
ctx.clear(); ctx.sync(); t0 = time.perf_counter()
ctx.render(1, 64 * 64, 1)
ctx.sync(); job = (time.perf_counter() - t0) / 64 * 1e3
tm = ctx.timing()
print("single-launch %%.3f ms | 56 calls pipelined %%.3f ms/step | one 64-window job %%.3f ms/step (render kernel %%.1f ms, accumulate %%.1f ms, %%d launches)" %% (np.median(ts[1:]), thr, job, tm["render_kernel_ms"], tm["resolve_kernel_ms"], tm["render_launches"]))
''' % REPO
    scene = args[0] if args else "bunny_scene.xml"
    kind = args[1] if len(args) > 1 else "0"
    W = args[2] if len(args) > 2 else "1280"
    H = args[3] if len(args) > 3 else "720"
    for f in sorted(os.listdir(VDIR)):
        if not f.endswith(".so"): continue
        env = dict(os.environ, CRT_LIB_PATH=os.path.join(VDIR, f), GPU_MAX_HW_QUEUES=os.environ.get("GPU_MAX_HW_QUEUES", "16"))
        r = subprocess.run([sys.executable, "-c", child, scene, kind, W, H], env=env, capture_output=True, text=True)
        print("%-40s %s %s" % (f, r.stdout.strip(), r.stderr.strip()[-200:] if r.returncode else ""))

if __name__ == "__main__":
    if sys.argv[1] == "build": build(sys.argv[2:])
    elif sys.argv[1] == "fastbuild": fastbuild(sys.argv[2:])
    else: run(sys.argv[2:])
