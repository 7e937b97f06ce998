"""
cpu_ray_tracer_amd — ctypes binding of libcrt_amd.so (the MI355X path-tracing back end).

This module is plumbing only: every call goes straight to the C ABI declared in include/crt_abi.h and
include/crt_host.h.  There is no Python or CPU implementation of the path behind it — if the shared library is
missing, or no HIP device is visible, the calls raise.

The directory is named `cpu-ray-tracer_amd` (not an importable identifier); load it with
    importlib.util.spec_from_file_location("cpu_ray_tracer_amd", ".../cpu-ray-tracer_amd/__init__.py")
as __graft_entry__.py, bench.py and tests/conftest.py do.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
LIB_PATH = os.environ.get("CRT_LIB_PATH") or os.path.join(HERE, "libcrt_amd.so")   # CRT_LIB_PATH: A/B builds of the same library (tools/ab_bench.py)

SCENE_FILE, SCENE_TLAS = 0, 1
UPDATE_TRANSFORMS, UPDATE_BOUNDS = 1, 2        # crt_update_scene flags
ACCEL_KDTREE, ACCEL_GRID = 1, 2                 # FileScene's alternative accelerators (crt_upload_alt_accel / crt_find_nearest_alt)
KD_NODE_DTYPE = np.dtype([("aabbMin", "<f4", 3), ("left", "<i4"), ("aabbMax", "<f4", 3), ("right", "<i4"), ("splitDistance", "<f4"), ("splitAxis", "<i4"),
                          ("firstTri", "<u4"), ("triCount", "<u4")])      # crt_kd_node


class CrtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("crt error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("depthLimit", C.c_int32), ("device", C.c_int32),
                ("tileFirst", C.c_int32), ("tileStride", C.c_int32), ("tileCount", C.c_int32),
                ("maxFramesPerLaunch", C.c_int32), ("collectStats", C.c_int32), ("renderStreams", C.c_int32)]


# ---- reference record layouts of include/crt_abi.h (ctypes mirrors; used by Context.upload_desc = INTEGRATION.md path A from Python) ----
class BvhS(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("nodesUsed", C.c_uint32), ("triangles", C.c_void_p), ("triCount", C.c_uint32), ("triangleIndices", C.c_void_p),
                ("objIdx", C.c_int32), ("matIdx", C.c_int32), ("T", C.c_float * 16), ("invT", C.c_float * 16)]


class TextureS(C.Structure):
    _fields_ = [("pixels", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


class MaterialS(C.Structure):
    _fields_ = [("reflectivity", C.c_float), ("refractivity", C.c_float), ("absorption", C.c_float * 3), ("texture", C.c_int32)]


class SceneDescS(C.Structure):
    _fields_ = [("kind", C.c_int32), ("bvhs", C.POINTER(BvhS)), ("bvhCount", C.c_uint32), ("tlasNodes", C.c_void_p), ("tlasNodeCount", C.c_uint32),
                ("objMatIdx", C.c_void_p), ("objCount", C.c_uint32), ("materials", C.POINTER(MaterialS)), ("materialCount", C.c_uint32),
                ("textures", C.POINTER(TextureS)), ("textureCount", C.c_uint32), ("floorTexture", C.c_int32), ("skyTexture", C.c_int32),
                ("lightT", C.c_float * 16), ("lightInvT", C.c_float * 16), ("lightSize", C.c_float),
                ("floorN", C.c_float * 3), ("floorD", C.c_float), ("floorInvto", C.c_float)]


class CountersS(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "primary", "interior_iters", "leaf_iters", "tri_tests", "tlas_iters", "blas_visits", "mesh_hits")]


class TimingS(C.Structure):
    _fields_ = [("render_kernel_ms", C.c_float), ("resolve_kernel_ms", C.c_float), ("render_launches", C.c_uint32), ("pool_launches", C.c_uint32), ("split_launches", C.c_uint32), ("reserved", C.c_uint32)]


TRI_DTYPE = np.dtype([("vertex0", "<f4", 3), ("vertex1", "<f4", 3), ("vertex2", "<f4", 3),
                      ("normal0", "<f4", 3), ("normal1", "<f4", 3), ("normal2", "<f4", 3),
                      ("uv0", "<f4", 2), ("uv1", "<f4", 2), ("uv2", "<f4", 2), ("centroid", "<f4", 3), ("objIdx", "<i4")])
NODE_DTYPE = np.dtype([("aabbMin", "<f4", 3), ("aabbMax", "<f4", 3), ("leftFirst", "<u4"), ("triCount", "<u4")])
TLAS_DTYPE = np.dtype([("aabbMin", "<f4", 3), ("leftRight", "<u4"), ("aabbMax", "<f4", 3), ("BLAS", "<u4")])
RAY_DTYPE = np.dtype([("O", "<f4", 3), ("D", "<f4", 3), ("inside", "<i4")])
HIT_DTYPE = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("objIdx", "<i4"), ("triIdx", "<i4"), ("traversed", "<i4"), ("tested", "<i4")])

# every symbol include/crt_abi.h and include/crt_host.h declare (tests check the library exports all of them)
ABI_SYMBOLS = ["crt_upload_primitive_scene", "crt_set_render_accel", "crt_upload_alt_accel", "crt_find_nearest_alt", "crt_update_scene", "crt_abi_version", "crt_device_count", "crt_create", "crt_destroy", "crt_last_error", "crt_upload_scene", "crt_set_camera",
               "crt_render", "crt_reserve", "crt_whitted_tick", "crt_sync", "crt_clear", "crt_read_accumulator", "crt_resolve_screen", "crt_find_nearest", "crt_get_counters",
               "crt_reset_counters", "crt_get_timing", "crt_get_tile_clocks", "crt_bind_accumulator", "crt_accumulator_device_ptr"]
HOST_SYMBOLS = ["crt_host_primitive_scene_create", "crt_host_primitive_scene_free", "crt_host_primitive_scene_set_time", "crt_host_primitive_scene_desc", "crt_host_primitive_scene_upload", "crt_host_scene_build_alt", "crt_host_scene_upload_alt", "crt_host_scene_alt_info", "crt_host_scene_alt_copy", "crt_host_scene_set_transform", "crt_host_scene_update", "crt_host_math_probe", "crt_host_vertex_dedup", "crt_host_last_error", "crt_host_scene_load", "crt_host_scene_free", "crt_host_scene_upload", "crt_host_scene_kind",
                "crt_host_scene_triangle_count", "crt_host_scene_bvh_count", "crt_host_scene_bvh_info", "crt_host_scene_bvh_copy",
                "crt_host_scene_bvh_move_and_refit", "crt_host_scene_blas_transform", "crt_host_scene_tlas_copy", "crt_host_camera_state", "crt_host_renderer_create",
                "crt_host_renderer_destroy", "crt_host_renderer_init", "crt_host_renderer_set_camera", "crt_host_renderer_set_passes",
                "crt_host_renderer_clear", "crt_host_renderer_tick", "crt_host_renderer_render", "crt_host_renderer_tick_whitted", "crt_host_renderer_spp",
                "crt_host_renderer_energy", "crt_host_renderer_accumulator", "crt_host_renderer_screen", "crt_host_renderer_ctx",
                "crt_host_obj_load", "crt_host_image_load", "crt_host_free"]


def build(force=False):
    """Compile libcrt_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    srcs = [os.path.join(HERE, "Makefile")]
    for root, _, files in os.walk(os.path.join(HERE, "csrc")):
        srcs += [os.path.join(root, f) for f in files]
    srcs += [os.path.join(REPO, "include", f) for f in ("crt_abi.h", "crt_host.h")]
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", HERE] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (there is no fallback path)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        # the library's test / diagnostic environment switches (CRT_RENDER_KERNEL, CRT_LAT_*, ...) are dead unless the process opts in; the test suite and the
        # tools do (tests/conftest.py, tools/*.py set CRT_ENABLE_DEBUG_HOOKS=1 for themselves), bench.py and a host application do not
        if os.environ.get("CRT_ENABLE_DEBUG_HOOKS") == "1":
            L.crt_debug_enable_hooks(1)
        L.crt_last_error.restype = C.c_char_p
        L.crt_last_error.argtypes = [C.c_void_p]
        L.crt_host_last_error.restype = C.c_char_p
        L.crt_host_renderer_energy.restype = C.c_float
        L.crt_host_renderer_accumulator.restype = C.POINTER(C.c_float)
        L.crt_host_renderer_screen.restype = C.POINTER(C.c_uint32)
        L.crt_host_renderer_ctx.restype = C.c_void_p
        L.crt_destroy.argtypes = [C.c_void_p]
        L.crt_host_scene_free.argtypes = [C.c_void_p]
        L.crt_host_renderer_destroy.argtypes = [C.c_void_p]
        L.crt_host_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def device_count():
    return int(lib().crt_device_count())


class Context:
    """crt_ctx: one device, one image (or a strided subset of its 16x16 tiles)."""

    def __init__(self, width, height, depth_limit=5, device=0, tile_first=0, tile_stride=1, tile_count=-1,
                 max_frames_per_launch=0, collect_stats=False, render_streams=0):
        self.L = lib()
        cfg = Config(width, height, depth_limit, device, tile_first, tile_stride, tile_count, max_frames_per_launch, int(collect_stats), render_streams)
        h = C.c_void_p()
        rc = self.L.crt_create(C.byref(h), C.byref(cfg))
        if rc != 0:
            raise CrtError(rc, self.L.crt_last_error(None).decode())
        self.h = h
        self.W, self.H = width, height
        self._owned = True

    @classmethod
    def borrow(cls, handle, width, height):
        o = cls.__new__(cls)
        o.L = lib()
        o.h = C.c_void_p(handle)
        o.W, o.H = width, height
        o._owned = False
        return o

    def close(self):
        if getattr(self, "h", None) and self._owned:
            self.L.crt_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise CrtError(rc, self.L.crt_last_error(self.h).decode())

    def set_camera(self, cam_pos, top_left, top_right, bottom_left):
        self._ck(self.L.crt_set_camera(self.h, _f3(cam_pos), _f3(top_left), _f3(top_right), _f3(bottom_left)))

    def set_camera_state(self, position, target):
        a = [(C.c_float * 3)() for _ in range(4)]
        rc = self.L.crt_host_camera_state(self.W, self.H, _f3(position), _f3(target), *a)
        if rc != 0:
            raise CrtError(rc, self.L.crt_host_last_error().decode())
        self._ck(self.L.crt_set_camera(self.h, *a))

    def render(self, spp_first, frames, passes=1):
        self._ck(self.L.crt_render(self.h, C.c_uint32(spp_first), C.c_uint32(frames), C.c_uint32(passes)))

    def reserve(self, frames, passes=1):
        self._ck(self.L.crt_reserve(self.h, C.c_uint32(frames), C.c_uint32(passes)))

    def whitted_tick(self):
        px = np.empty((self.H, self.W), np.uint32)
        self._ck(self.L.crt_whitted_tick(self.h, _p(px)))
        return px

    def sync(self):
        self._ck(self.L.crt_sync(self.h))

    def clear(self):
        self._ck(self.L.crt_clear(self.h))

    def accumulator(self):
        a = np.empty((self.H, self.W, 4), np.float32)
        self._ck(self.L.crt_read_accumulator(self.h, _p(a)))
        return a

    def resolve_screen(self, scale):
        px = np.empty((self.H, self.W), np.uint32)
        e = C.c_float()
        self._ck(self.L.crt_resolve_screen(self.h, C.c_float(scale), _p(px), C.byref(e)))
        return px, e.value

    def set_render_accel(self, kind):
        """crt_set_render_accel: 0 = BVH / TLAS, ACCEL_KDTREE / ACCEL_GRID = Sample and Trace go through the uploaded alternative accelerator"""
        self._ck(self.L.crt_set_render_accel(self.h, int(kind)))

    def find_nearest_alt(self, kind, O, D):
        """scene.FindNearest with FileScene's KD-tree / grid in place of the BVH (crt_find_nearest_alt)"""
        O = np.ascontiguousarray(O, np.float32).reshape(-1, 3); D = np.ascontiguousarray(D, np.float32).reshape(-1, 3)
        rays = np.zeros(O.shape[0], RAY_DTYPE); rays["O"] = O; rays["D"] = D
        hits = np.zeros(O.shape[0], HIT_DTYPE)
        self._ck(self.L.crt_find_nearest_alt(self.h, int(kind), _p(rays), _p(hits), C.c_size_t(O.shape[0])))
        return hits

    def upload_desc(self, kind, bvhs, textures, floor_texture, sky_texture, materials, light_T, light_invT, light_size=0.5,
                    floor_n=(0, 1, 0), floor_d=1.0, floor_invto=None, obj_mat_idx=None, tlas_nodes=None, update_what=None):
        """crt_upload_scene with arrays BUILT ELSEWHERE, in the reference's own layouts (INTEGRATION.md path A): bvhs = dicts with `nodes` (32-byte BVHNode
        records), `tris` (112-byte Tri records), `triIndices` (uint32) and, for two-level scenes, objIdx / matIdx / T / invT; textures = uint32 (h, w)
        arrays of 0x00RRGGBB texels; materials = (reflectivity, refractivity, absorption[3], texture index or -1).  Nothing of the repo's host front is involved."""
        keep = []
        bs = (BvhS * len(bvhs))()
        for i, b in enumerate(bvhs):
            nodes = np.ascontiguousarray(b["nodes"]); tris = np.ascontiguousarray(b["tris"]); idx = np.ascontiguousarray(b["triIndices"], np.uint32)
            assert nodes.dtype.itemsize == 32 and tris.dtype.itemsize == 112
            keep += [nodes, tris, idx]
            bs[i].nodes = nodes.ctypes.data; bs[i].nodesUsed = int(b.get("nodesUsed", len(nodes))); bs[i].triangles = tris.ctypes.data; bs[i].triCount = len(tris)
            bs[i].triangleIndices = idx.ctypes.data; bs[i].objIdx = int(b.get("objIdx", -1)); bs[i].matIdx = int(b.get("matIdx", -1))
            bs[i].T[:] = list(np.asarray(b.get("T", np.eye(4)), np.float32).reshape(16)); bs[i].invT[:] = list(np.asarray(b.get("invT", np.eye(4)), np.float32).reshape(16))
        ts = (TextureS * len(textures))()
        for i, t in enumerate(textures):
            t = np.ascontiguousarray(t, np.uint32); keep.append(t)
            ts[i].pixels = t.ctypes.data; ts[i].height, ts[i].width = t.shape
        ms = (MaterialS * max(len(materials), 1))()
        for i, (refl, refr, ab, tex) in enumerate(materials):
            ms[i].reflectivity = refl; ms[i].refractivity = refr; ms[i].absorption[:] = list(ab); ms[i].texture = tex
        d = SceneDescS()
        d.kind = kind; d.bvhs = bs; d.bvhCount = len(bvhs)
        if tlas_nodes is not None:
            tn = np.ascontiguousarray(tlas_nodes); keep.append(tn); d.tlasNodes = tn.ctypes.data; d.tlasNodeCount = len(tn)
        if obj_mat_idx is not None:
            om = np.ascontiguousarray(obj_mat_idx, np.int32); keep.append(om); d.objMatIdx = om.ctypes.data; d.objCount = len(om)
        d.materials = ms; d.materialCount = len(materials); d.textures = ts; d.textureCount = len(textures)
        d.floorTexture = floor_texture; d.skyTexture = sky_texture
        d.lightT[:] = list(np.asarray(light_T, np.float32).reshape(16)); d.lightInvT[:] = list(np.asarray(light_invT, np.float32).reshape(16)); d.lightSize = light_size
        d.floorN[:] = list(floor_n); d.floorD = floor_d
        if floor_invto is None:                                              # file_scene.cpp:16: Plane(.., texW / 100) with an integer division
            floor_invto = 1.0 / float(max(int(textures[floor_texture].shape[1]) // 100, 1)) if 0 <= floor_texture < len(textures) else 1.0
        d.floorInvto = floor_invto
        if update_what is not None:                                          # the same description to crt_update_scene (in-place update of the uploaded scene)
            self._ck(self.L.crt_update_scene(self.h, C.byref(d), C.c_uint32(update_what)))
            return
        self._ck(self.L.crt_upload_scene(self.h, C.byref(d)))

    def find_nearest(self, O, D, inside=None):
        O = np.asarray(O, np.float32).reshape(-1, 3)
        D = np.asarray(D, np.float32).reshape(-1, 3)
        rays = np.zeros(O.shape[0], RAY_DTYPE)
        rays["O"], rays["D"] = O, D
        if inside is not None:
            rays["inside"] = inside
        hits = np.zeros(O.shape[0], HIT_DTYPE)
        self._ck(self.L.crt_find_nearest(self.h, _p(rays), _p(hits), C.c_size_t(O.shape[0])))
        return hits

    def counters(self):
        c = CountersS()
        self._ck(self.L.crt_get_counters(self.h, C.byref(c)))
        return {n: int(getattr(c, n)) for n, _ in c._fields_}

    def reset_counters(self):
        self._ck(self.L.crt_reset_counters(self.h))

    def timing(self):
        t = TimingS()
        self._ck(self.L.crt_get_timing(self.h, C.byref(t)))
        return dict(render_kernel_ms=t.render_kernel_ms, resolve_kernel_ms=t.resolve_kernel_ms, render_launches=t.render_launches, pool_launches=t.pool_launches, split_launches=t.split_launches)

    def tile_clocks(self, tile_count):
        a = np.zeros((tile_count, 2), np.uint64)
        self._ck(self.L.crt_get_tile_clocks(self.h, _p(a)))
        return a

    def bind_accumulator(self, device_ptr):
        self._ck(self.L.crt_bind_accumulator(self.h, C.c_void_p(device_ptr)))

    def accumulator_device_ptr(self):
        p = C.c_void_p()
        self._ck(self.L.crt_accumulator_device_ptr(self.h, C.byref(p)))
        return p.value


class HostScene:
    """FileScene / TLASFileScene built on the CPU by the C++ host front (XML + OBJ + textures + SAH-BVH / TLAS)."""

    def __init__(self, xml_path, kind, base_dir=None):
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.crt_host_scene_load(xml_path.encode(), int(kind), (base_dir or "").encode(), C.byref(h))
        if rc != 0:
            raise CrtError(rc, self.L.crt_host_last_error().decode())
        self.h = h
        self.kind = int(kind)

    def close(self):
        if getattr(self, "h", None):
            self.L.crt_host_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise CrtError(rc, self.L.crt_host_last_error().decode())

    def upload(self, ctx):
        self._ck(self.L.crt_host_scene_upload(self.h, ctx.h))

    def triangle_count(self):
        return self.L.crt_host_scene_triangle_count(self.h)

    def bvh_count(self):
        return self.L.crt_host_scene_bvh_count(self.h)

    def bvh(self, i=0):
        nu, tc, md = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._ck(self.L.crt_host_scene_bvh_info(self.h, i, C.byref(nu), C.byref(tc), C.byref(md)))
        nodes = np.zeros(nu.value, NODE_DTYPE)
        idx = np.zeros(tc.value, np.uint32)
        tris = np.zeros(tc.value, TRI_DTYPE)
        self._ck(self.L.crt_host_scene_bvh_copy(self.h, i, _p(nodes), _p(idx), _p(tris)))
        return dict(nodes=nodes, triIndices=idx, tris=tris, nodesUsed=nu.value, maxDepth=md.value)

    def move_and_refit(self, i, positions):
        """BVH::Refit for moved vertices: positions = (triCount, 3, 3) floats in the reference's triangle order; upload() again afterwards"""
        positions = np.ascontiguousarray(positions, np.float32)
        self._ck(self.L.crt_host_scene_bvh_move_and_refit(self.h, int(i), _p(positions), C.c_uint32(positions.shape[0])))

    def build_alt(self, kind):
        """KDTree::Build / Grid::Build over the FileScene's triangles on the host; returns the flattened structure (the layout crt_upload_alt_accel takes)"""
        self._ck(self.L.crt_host_scene_build_alt(self.h, int(kind)))
        info = (C.c_uint32 * 4)()
        self._ck(self.L.crt_host_scene_alt_info(self.h, int(kind), info))
        if kind == ACCEL_KDTREE:
            nodes = np.zeros(info[0], KD_NODE_DTYPE); refs = np.zeros(max(info[1], 1), np.uint32)
            self._ck(self.L.crt_host_scene_alt_copy(self.h, int(kind), _p(nodes), _p(refs), None))
            return dict(nodes=nodes, refs=refs[:info[1]], maxDepth=int(info[2]), nodesUsed=int(info[3]))
        res = np.array([info[0], info[1], info[2]], np.int32)
        start = np.zeros(int(res.prod()) + 1, np.uint32); refs = np.zeros(max(info[3], 1), np.int32); f = np.zeros(9, np.float32)
        self._ck(self.L.crt_host_scene_alt_copy(self.h, int(kind), _p(start), _p(refs), _p(f)))
        return dict(resolution=res, cellSize=f[0:3].copy(), boundsMin=f[3:6].copy(), boundsMax=f[6:9].copy(), cellStart=start, refs=refs[:info[3]])

    def upload_alt(self, ctx, kind):
        self._ck(self.L.crt_host_scene_upload_alt(self.h, ctx.h, int(kind)))

    def set_transform(self, i, T):
        """BLASBVH::SetTransform(T) of instance i (T = 4x4 row-major, rigid) + TLASBVH::Build on the host; update(ctx, UPDATE_TRANSFORMS) moves it to the device"""
        T = np.ascontiguousarray(T, np.float32).reshape(16)
        self._ck(self.L.crt_host_scene_set_transform(self.h, int(i), _p(T)))

    def update(self, ctx, what):
        """in-place device update (crt_update_scene): UPDATE_TRANSFORMS after set_transform, UPDATE_BOUNDS after move_and_refit; no re-upload"""
        self._ck(self.L.crt_host_scene_update(self.h, ctx.h, C.c_uint32(what)))

    def blas_transform(self, i):
        T, invT, lo, hi = np.zeros(16, np.float32), np.zeros(16, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
        self._ck(self.L.crt_host_scene_blas_transform(self.h, i, _p(T), _p(invT), _p(lo), _p(hi)))
        return T, invT, lo, hi

    def tlas(self):
        nodes = np.zeros(2 * self.bvh_count(), TLAS_DTYPE)
        nu = C.c_uint32()
        self._ck(self.L.crt_host_scene_tlas_copy(self.h, _p(nodes), C.byref(nu)))
        return nodes, nu.value


class PrimitiveSceneS(C.Structure):
    _fields_ = [("quadT", C.c_float * 16), ("quadInvT", C.c_float * 16), ("quadSize", C.c_float), ("spherePos", C.c_float * 3),
                ("cubeMin", C.c_float * 3), ("cubeMax", C.c_float * 3), ("cubeM", C.c_float * 16), ("cubeInvM", C.c_float * 16),
                ("torusT", C.c_float * 16), ("torusInvT", C.c_float * 16), ("torusRt2", C.c_float), ("torusRc2", C.c_float), ("torusR2", C.c_float),
                ("reflectivity", C.c_float * 11), ("refractivity", C.c_float * 11), ("absorption", C.c_float * 33), ("red", TextureS), ("blue", TextureS)]


class HostPrimitiveScene:
    """PrimitiveScene of the reference (infra/scene/primitive_scene.cpp): constructor, SetTime, upload through crt_upload_primitive_scene"""

    def __init__(self, assets_dir=None):
        self.L = lib()
        h = C.c_void_p()
        r = self.L.crt_host_primitive_scene_create(assets_dir.encode() if assets_dir else None, C.byref(h))
        if r != 0:
            raise CrtError(r, self.L.crt_host_last_error().decode())
        self.h = h

    def set_time(self, t):
        self.L.crt_host_primitive_scene_set_time(self.h, C.c_float(t))

    def desc(self):
        d = PrimitiveSceneS()
        self.L.crt_host_primitive_scene_desc(self.h, C.byref(d))
        return d

    def state(self):
        """the 108 floats of the oracle's orc_prim_state: quad T, invT, cube M, invM, torus T, invT, sphere position, rt2, rc2, r2, cube box"""
        d = self.desc()
        return np.array(list(d.quadT) + list(d.quadInvT) + list(d.cubeM) + list(d.cubeInvM) + list(d.torusT) + list(d.torusInvT) + list(d.spherePos)
                        + [d.torusRt2, d.torusRc2, d.torusR2] + list(d.cubeMin) + list(d.cubeMax), np.float32)

    def upload(self, ctx):
        r = self.L.crt_host_primitive_scene_upload(self.h, ctx.h)
        if r != 0:
            raise CrtError(r, self.L.crt_last_error(ctx.h).decode())

    def close(self):
        if self.h:
            self.L.crt_host_primitive_scene_free(self.h); self.h = None


class HostRenderer:
    """Renderer facade: Init / Tick / ClearAccumulator with the reference's public members."""

    def __init__(self, scene, width, height, device=0):
        self.L = lib()
        self.scene = scene
        h = C.c_void_p()
        rc = self.L.crt_host_renderer_create(scene.h, width, height, device, C.byref(h))
        if rc != 0:
            raise CrtError(rc, self.L.crt_host_last_error().decode())
        self.h = h
        self.W, self.H = width, height

    def close(self):
        if getattr(self, "h", None):
            self.L.crt_host_renderer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise CrtError(rc, self.L.crt_host_last_error().decode())

    def init(self):
        self._ck(self.L.crt_host_renderer_init(self.h))

    def set_camera(self, position, target):
        self._ck(self.L.crt_host_renderer_set_camera(self.h, _f3(position), _f3(target)))

    def set_passes(self, passes):
        self._ck(self.L.crt_host_renderer_set_passes(self.h, passes))

    def clear(self):
        self._ck(self.L.crt_host_renderer_clear(self.h))

    def tick(self, dt=0.0):
        self._ck(self.L.crt_host_renderer_tick(self.h, C.c_float(dt)))

    def render(self, frames):
        self._ck(self.L.crt_host_renderer_render(self.h, frames))

    def tick_whitted(self):
        self._ck(self.L.crt_host_renderer_tick_whitted(self.h))

    @property
    def spp(self):
        return self.L.crt_host_renderer_spp(self.h)

    @property
    def energy(self):
        return float(self.L.crt_host_renderer_energy(self.h))

    def accumulator(self):
        p = self.L.crt_host_renderer_accumulator(self.h)
        return np.ctypeslib.as_array(p, shape=(self.H, self.W, 4)).copy()

    def screen(self):
        p = self.L.crt_host_renderer_screen(self.h)
        return np.ctypeslib.as_array(p, shape=(self.H, self.W)).copy()

    def context(self):
        return Context.borrow(self.L.crt_host_renderer_ctx(self.h), self.W, self.H)


def load_obj(path):
    L = lib()
    n = C.c_uint32()
    pp, pn, pu = C.POINTER(C.c_float)(), C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
    rc = L.crt_host_obj_load(path.encode(), C.byref(n), C.byref(pp), C.byref(pn), C.byref(pu))
    if rc != 0:
        raise CrtError(rc, L.crt_host_last_error().decode())
    pos = np.ctypeslib.as_array(pp, shape=(n.value, 3)).copy()
    nrm = np.ctypeslib.as_array(pn, shape=(n.value, 3)).copy()
    uv = np.ctypeslib.as_array(pu, shape=(n.value, 2)).copy()
    for q in (pp, pn, pu):
        L.crt_host_free(C.cast(q, C.c_void_p))
    return pos, nrm, uv


def load_image(path):
    L = lib()
    w, h = C.c_int(), C.c_int()
    px = C.POINTER(C.c_uint32)()
    rc = L.crt_host_image_load(path.encode(), C.byref(w), C.byref(h), C.byref(px))
    if rc != 0:
        raise CrtError(rc, L.crt_host_last_error().decode())
    a = np.ctypeslib.as_array(px, shape=(h.value, w.value)).copy()
    L.crt_host_free(C.cast(px, C.c_void_p))
    return a


# ------------------------------------------------------------------------------------------------------------
# multi-GPU work split (one process per GPU, torch.distributed over RCCL) — pure arithmetic, shared by bench.py and tests
# ------------------------------------------------------------------------------------------------------------
def host_math_probe(inputs):
    """test entry: the host front's tmplmath.h restatements (csrc/host/hmath.h), layout of the real-reference harness's ref_math_probe (tests/golden/make_golden.py)"""
    inputs = np.ascontiguousarray(inputs, np.float32).reshape(-1, 12)
    out = np.zeros((len(inputs), 120), np.float32)
    lib().crt_host_math_probe(inputs.ctypes.data_as(C.c_void_p), C.c_uint32(len(inputs)), out.ctypes.data_as(C.c_void_p))
    return out


def host_vertex_dedup(v8):
    """test entry: the host front's unique-vertex table (csrc/host/accel.cpp Dedup): (corner indices, unique vertices)"""
    v8 = np.ascontiguousarray(v8, np.float32).reshape(-1, 8)
    idx = np.zeros(len(v8), np.uint32); uniq = np.zeros((len(v8), 8), np.float32)
    L = lib(); L.crt_host_vertex_dedup.restype = C.c_uint32
    n = L.crt_host_vertex_dedup(v8.ctypes.data_as(C.c_void_p), C.c_uint32(len(v8)), idx.ctypes.data_as(C.c_void_p), uniq.ctypes.data_as(C.c_void_p))
    return idx, uniq[:n].copy()


def spp_window(rank, frames_per_rank, first_spp=1):
    """Weak scaling: rank r renders frames whose spp counter runs first_spp + r*F .. first_spp + (r+1)*F - 1.
    (tile, frame) streams are independent (renderer.cpp:120), so windows can be rendered anywhere and summed."""
    return first_spp + rank * frames_per_rank


def tile_partition(rank, world, n_tiles):
    """Strong scaling: rank r owns tiles r, r + world, r + 2*world, ... (interleaved: heavy image regions are shared out).
    Returns (tileFirst, tileStride, tileCount) for crt_config.  Each pixel is non-zero on exactly one rank, so a sum
    all-reduce of the accumulators reproduces the single-GPU image exactly."""
    count = (n_tiles - rank + world - 1) // world if rank < n_tiles else 0
    return rank, world, count


def reduce_accumulator(tensor, dist, dst=0):
    """ncclReduce of the float4 accumulators to one rank: only rank `dst` ends up with the whole image.  With tile ownership
    every pixel is non-zero on exactly one rank, so the sum is exact (x + 0 ...)."""
    dist.reduce(tensor, dst=dst, op=dist.ReduceOp.SUM)
    return tensor


def allreduce_accumulator(tensor, dist):
    """One collective per step: sum of the float4 accumulators over all ranks (RCCL over xGMI on GPUs, gloo in CPU tests)."""
    dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor
