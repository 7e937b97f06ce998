"""CPU tests of the product's host side: the C-ABI library loads and exports every declared symbol, the C++ host front
(own OBJ / XML / image loaders + CPU SAH-BVH / TLAS build) reproduces the oracle's structures bit for bit, and errors come
back as codes.  No compute call is made (there is no GPU here and the library has no CPU path)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ASSETS, REPO, scene_path

SCENES = [("bunny_scene.xml", 0), ("cube_scene.xml", 0), ("tlas_scene.xml", 1), ("tlas_scene.xml", 0), ("tower_scene.xml", 0)]


def test_library_exports_every_declared_symbol(crt):
    lib = crt.lib()
    declared = []
    for hdr in ("crt_abi.h", "crt_host.h"):
        text = open(os.path.join(REPO, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared += re.findall(r"\b(crt_[a-z_0-9]+)\s*\(", text)
    declared = sorted(set(declared))
    assert len(declared) >= 40
    for sym in declared:
        assert hasattr(lib, sym), "include/*.h declares %s but libcrt_amd.so does not export it" % sym
    assert set(crt.ABI_SYMBOLS + crt.HOST_SYMBOLS) == set(declared)
    assert lib.crt_abi_version() == 3


def test_record_layouts(crt):
    assert crt.TRI_DTYPE.itemsize == 112 and crt.NODE_DTYPE.itemsize == 32 and crt.TLAS_DTYPE.itemsize == 32
    assert C.sizeof(crt.Config) == 40 and C.sizeof(crt.CountersS) == 64


@pytest.mark.parametrize("xml,kind", SCENES)
def test_host_build_matches_oracle(crt, orc, xml, kind):
    hs = crt.HostScene(scene_path(xml), kind, ASSETS)
    o, _ = orc.load_scene(scene_path(xml), kind, ASSETS)
    assert hs.bvh_count() == o.bvh_count()
    for i in range(hs.bvh_count()):
        a, b = hs.bvh(i), o.bvh(i)
        assert (a["nodesUsed"], a["maxDepth"]) == (b["nodesUsed"], b["maxDepth"])
        for k in ("nodes", "triIndices", "tris"):
            assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), (i, k)
        if kind == 1:
            for x, y in zip(hs.blas_transform(i), o.blas_transform(i)):
                assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    if kind == 1:
        (na, ua), (nb, ub) = hs.tlas(), o.tlas()
        assert ua == ub and np.array_equal(na.view(np.uint8), nb.view(np.uint8))


def test_loaders_match_independent_readers(crt, orc):
    for m in ("cube", "bunny", "teapot", "wok", "japanese_torii_gate", "watch-tower", "log_fence"):
        p = os.path.join(ASSETS, m + ".obj")
        for a, b in zip(crt.load_obj(p), orc.read_obj(p)):
            assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), m
    for f in ("textures/Stylized_Pavement_basecolor.png", "textures/Stylized_Wood_basecolor.tga", "textures/Defuse_wok.png", "textures/Wood_Tower_Col.png", "sky_gradient.png"):
        p = os.path.join(ASSETS, f)
        assert np.array_equal(crt.load_image(p), orc.pack_rgb(orc.read_image(p))), f


@pytest.mark.parametrize("xml,kind", [("bunny_scene.xml", 0), ("tlas_scene.xml", 1)])
def test_refit_matches_oracle(crt, orc, xml, kind):
    """BVH::Refit / BLASBVH::Refit for moved vertices in the host front (bvh.cpp:26-43): node arrays, and for the two-level scene the
    instance world bounds + rebuilt TLAS, bit-identical with the oracle's (which is pinned to the real reference's Refit)."""
    from test_oracle_pinning import deform
    hs = crt.HostScene(scene_path(xml), kind, ASSETS)
    o, _ = orc.load_scene(scene_path(xml), kind, ASSETS)
    i = hs.bvh_count() - 1
    t = hs.bvh(i)["tris"]
    moved = deform(np.stack([t["vertex0"], t["vertex1"], t["vertex2"]], axis=1))
    before = hs.bvh(i)["nodes"].copy()
    hs.move_and_refit(i, moved)
    o.move_and_refit(i, moved)
    a, b = hs.bvh(i), o.bvh(i)
    assert not np.array_equal(a["nodes"].view(np.uint8), before.view(np.uint8))
    assert np.array_equal(a["nodes"].view(np.uint8), b["nodes"].view(np.uint8)) and np.array_equal(a["tris"].view(np.uint8), b["tris"].view(np.uint8))
    assert np.array_equal(a["triIndices"], b["triIndices"])
    if kind == 1:
        for x, y in zip(hs.blas_transform(i), o.blas_transform(i)):
            assert np.array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32))
        (na, ua), (nb, ub) = hs.tlas(), o.tlas()
        assert ua == ub and np.array_equal(na.view(np.uint8), nb.view(np.uint8))
    with pytest.raises(crt.CrtError):
        hs.move_and_refit(i, moved[:-1])


def test_host_camera_and_textures_match_reference_golden(crt):
    """host front vs the REAL reference's outputs (tests/golden ref_camera / ref_texture, made by oracle/_ref): Camera::SetCameraState's
    frustum corners at the resolution template/camera.h is compiled for, and the texels of Texture::LoadFromFile for PNG / TGA / JPEG"""
    import json, zlib
    G = json.load(open(os.path.join(REPO, "tests", "golden", "golden.json")))
    crc = lambda a: int(zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff)
    for name, g in G["ref_camera"]["cams"].items():
        if g["pos_target"] is None:
            continue
        a = [(C.c_float * 3)() for _ in range(4)]
        pos, tgt = g["pos_target"]
        assert crt.lib().crt_host_camera_state(1024, 640, (C.c_float * 3)(*pos), (C.c_float * 3)(*tgt), *a) == 0
        assert crc(np.array([list(x) for x in a], np.float32)) == g["corners"], name
    t = G["ref_texture"]
    assert crc(crt.load_image(os.path.join(ASSETS, t["file"]))) == t["texels"]
    assert crc(crt.load_image(os.path.join(ASSETS, "textures", "Stylized_Wood_basecolor.tga"))) == t["tga"]
    assert crc(crt.load_image(os.path.join(ASSETS, "textures", "Wood_Tower_Col.jpg"))) == t["jpg"]


def test_jpeg_loader_matches_stb_golden(crt):
    """JPEG in the host loader (sequential and progressive): texels must be the ones the reference's stbi_load produces (template/texture.h:18).
    tests/golden/jpeg/*.jpg were decoded by the REAL lib/stb_image.h (oracle/_ref) when the fixtures were made
    (tests/golden/make_jpeg_golden.py): 4:4:4 / 4:2:2 / 4:2:0, greyscale, odd sizes down to 1x1, restart intervals, optimised
    Huffman tables, progressive files (spectral selection + successive approximation), and the reference's own Wood_Tower_Col.jpg
    (BASELINE config 4's texture)."""
    import json, zlib
    G = json.load(open(os.path.join(REPO, "tests", "golden", "jpeg_golden.json")))
    for name, g in sorted(G.items()):
        path = os.path.join(ASSETS, "textures", name + ".jpg") if name == "Wood_Tower_Col" else os.path.join(REPO, "tests", "golden", "jpeg", name + ".jpg")
        a = crt.load_image(path)
        assert list(a.shape) == g["shape"][:2], name
        assert (zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff) == g["packed"], name
    # the stb-decoded PNG fixture of the same texture (tools/make_tower_texture.py) holds the same texels
    assert np.array_equal(crt.load_image(os.path.join(ASSETS, "textures", "Wood_Tower_Col.jpg")), crt.load_image(os.path.join(ASSETS, "textures", "Wood_Tower_Col.png")))
    with pytest.raises(crt.CrtError) as e:
        crt.load_image(os.path.join(REPO, "tests", "golden", "jpeg", "cmyk_32x32.jpg"))
    assert "component" in str(e.value) and e.value.code == -6
    bad = open(os.path.join(REPO, "tests", "golden", "jpeg", "rgb420_16x16_q50.jpg"), "rb").read()
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        for cut in (3, 100, len(bad) // 2):
            p = os.path.join(d, "cut%d.jpg" % cut)
            open(p, "wb").write(bad[:cut])
            with pytest.raises(crt.CrtError):
                crt.load_image(p)


def test_obj_edge_cases(crt, orc, tmp_path):
    """ragged input: relative indices, missing vt/vn, quads on either diagonal, a pentagon, blank lines, CRLF, exponents."""
    text = "\r\n".join([
        "# comment", "", "v 0 0 0", "v 1 0 0", "v 1 1 0", "v 0 1 0", "v 0.5 1.5e0 0", "v 2 0 1", "v 2 1.0 -.5", "v +3 0 0",
        "vn 0 0 1", "vt 0.25 0.75", "o thing", "s off",
        "f 1 2 3", "f 1//1 2//1 3//1", "f 1/1 2/1 3/1", "f 1/1/1 2/1/1 3/1/1",
        "f -8 -7 -6", "f 1 2 3 4", "f 2 6 7 3", "f 1 2 3 5 4", "f 1 2", ""])
    p = tmp_path / "edge.obj"
    p.write_bytes(text.encode())
    a, b = crt.load_obj(str(p)), orc.read_obj(str(p))
    assert a[0].shape[0] == 3 * (5 + 2 + 2 + 3)
    for x, y in zip(a, b):
        assert x.shape == y.shape and np.array_equal(x.view(np.uint32), y.view(np.uint32))


def test_loader_errors_are_codes_not_crashes(crt, tmp_path):
    with pytest.raises(crt.CrtError) as e:
        crt.HostScene(str(tmp_path / "nope.xml"), 0, ASSETS)
    assert e.value.code == -6 and "nope.xml" in str(e.value)
    bad = tmp_path / "bad.xml"
    bad.write_text("<scene><scene_name>x</scene_name></scene>")
    with pytest.raises(crt.CrtError) as e:
        crt.HostScene(str(bad), 0, ASSETS)
    assert e.value.code == -6 and "light_position" in str(e.value)
    xml = open(scene_path("bunny_scene.xml")).read().replace("../assets/bunny.obj", "../assets/missing_model.obj")
    m = tmp_path / "missing_model.xml"
    m.write_text(xml)
    with pytest.raises(crt.CrtError) as e:
        crt.HostScene(str(m), 0, ASSETS)
    assert "missing_model.obj" in str(e.value)
    with pytest.raises(crt.CrtError):
        crt.load_obj(str(tmp_path / "absent.obj"))
    zero = tmp_path / "zero.obj"
    zero.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 1 2\n")          # index 0 is invalid in OBJ
    with pytest.raises(crt.CrtError):
        crt.load_obj(str(zero))
    with pytest.raises(crt.CrtError):
        crt.HostScene(scene_path("bunny_scene.xml"), 7, ASSETS)       # unknown scene kind


def test_material_index_out_of_range_is_rejected(crt, tmp_path):
    xml = open(scene_path("bunny_scene.xml")).read().replace("<material_idx>0</material_idx>", "<material_idx>3</material_idx>")
    p = tmp_path / "badmat.xml"
    p.write_text(xml)
    with pytest.raises(crt.CrtError) as e:
        crt.HostScene(str(p), 0, ASSETS)
    assert "material_idx" in str(e.value)


def test_camera_state_matches_oracle(crt, orc):
    o, _ = orc.load_scene(scene_path("cube_scene.xml"), 0, ASSETS)
    o.renderer_init(1280, 720)
    o.set_camera_state((1.5, 0.7, -3.0), (0.2, -0.1, 2.0))
    want = o.camera()
    got = [(C.c_float * 3)() for _ in range(4)]
    lib = crt.lib()
    assert lib.crt_host_camera_state(1280, 720, (C.c_float * 3)(1.5, 0.7, -3.0), (C.c_float * 3)(0.2, -0.1, 2.0), *got) == 0
    for g, w in zip(got, want):
        assert np.array_equal(np.array(list(g), np.float32).view(np.uint32), w.view(np.uint32))


def test_no_device_means_error_not_fallback(crt):
    """The product must fail loudly without a HIP device — never route through the oracle or any CPU path."""
    if crt.device_count() > 0:
        pytest.skip("a HIP device is visible")
    with pytest.raises(crt.CrtError) as e:
        crt.Context(64, 64)
    assert e.value.code == -3


def test_product_does_not_reference_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    pkg = os.path.join(REPO, "cpu-ray-tracer_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                text = open(os.path.join(root, f), errors="replace").read()
                assert "crt_oracle" not in text and "import orc" not in text and "oracle/" not in text.replace("oracle/_ref", ""), os.path.join(root, f)


def _plan(crt, cost, windows, frames, pool, window_ticks=0.0):
    L = crt.lib()
    L.crt_debug_plan_job.restype = C.c_int
    L.crt_debug_plan_job.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_double, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    n = len(cost)
    cost = np.ascontiguousarray(cost, np.uint32)
    table = np.zeros(n * windows * 64, np.uint32); order = np.zeros(n, np.uint32); head = C.c_uint32(0)
    k = L.crt_debug_plan_job(cost.ctypes.data, n, windows, frames, int(pool), float(window_ticks), table.ctypes.data, table.size, C.byref(head), order.ctypes.data)
    assert k >= 0, k
    t = table[:k]
    return dict(tile=t & 0xffff, base=(t >> 16) & 63, lanes=1 << ((t >> 22) & 7), window=t >> 25, head=head.value, order=order)


@pytest.mark.parametrize("seed,n,windows,frames,pool", [(1, 3600, 20, 1280, True), (2, 450, 20, 1280, False), (3, 3600, 5, 300, False), (4, 900, 3, 130, True),
                                                        (5, 8040, 7, 448, True), (6, 64, 2, 128, False), (7, 3600, 64, 4096, True)])
def test_job_planner_covers_every_stream_exactly_once(crt, seed, n, windows, frames, pool):
    """abi.cpp plan_job (host logic, no GPU): whatever the tile costs, the block table names every (tile, window, frame) of its tiles exactly once, its tiles are
    the leading ones of the cost order (the rest is the pool's), and a more expensive tile never gets wider wavefronts than a cheaper one."""
    rng = np.random.default_rng(seed)
    cost = (rng.pareto(1.2, n) * 2.0e4 + rng.integers(1000, 30000, n)).clip(0, 4.0e6).astype(np.uint32)     # heavy-tailed, like an image with one expensive object
    for window_ticks in (0.0, float(cost.sum()) / 4096.0):
        p = _plan(crt, cost, windows, frames, pool, window_ticks)
        order = p["order"]
        assert sorted(order.tolist()) == list(range(n)) and np.all(np.diff(cost[order].astype(np.int64)) <= 0)
        if len(p["tile"]) == 0:
            assert p["head"] == 0
            continue
        tiles = np.unique(p["tile"])
        expect = order[:p["head"]] if pool else order
        assert set(tiles.tolist()) == set(expect.tolist())
        if pool: assert 0 < p["head"] < n
        rank = np.empty(n, np.int64); rank[order] = np.arange(n)
        lanes_of = {}
        for t, b, l, w in zip(p["tile"].tolist(), p["base"].tolist(), p["lanes"].tolist(), p["window"].tolist()):
            assert w < windows and b % l == 0
            lanes_of.setdefault(t, l); assert lanes_of[t] == l                         # one width per tile
        # coverage: per (tile, window) the wavefronts tile [0, frames of the window) without overlap
        cover = np.zeros((n, windows, 64), np.int32)
        for t, b, l, w in zip(p["tile"].tolist(), p["base"].tolist(), p["lanes"].tolist(), p["window"].tolist()):
            cover[t, w, b:b + l] += 1
        for w in range(windows):
            fw = min(64, frames - 64 * w)
            assert np.all(cover[tiles, w, :fw] == 1), w
            if fw < 64: assert np.all(cover[tiles, w, ((fw + 63) // 64) * 64:] == 0)
        assert cover.max() == 1
        widths = np.array([lanes_of[t] for t in expect.tolist()])
        assert np.all(np.diff(widths) >= 0)                                               # along the cost order wavefronts only get wider


def test_job_planner_leaves_long_jobs_of_even_tiles_to_the_pool(crt):
    cost = np.full(3600, 5000, np.uint32)
    p = _plan(crt, cost, 64, 4096, True)
    assert len(p["tile"]) == 0 and p["head"] == 0


def test_latency_tuner_step_is_monotone_and_valid(crt):
    """abi.cpp next_lanes (host logic, no GPU): widths stay powers of two that divide 64; a tile over the aim never gets wider, a tile far below it never narrower;
    a higher cost never yields a wider wavefront than a lower one at the same width and aim."""
    L = crt.lib()
    L.crt_debug_next_lanes.restype = C.c_int
    L.crt_debug_next_lanes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_double, C.c_void_p]
    rng = np.random.default_rng(11)
    n = 4096
    lanes = rng.choice(np.array([64, 32, 16, 8, 4, 2, 1], np.uint8), n)
    cost = rng.integers(1, 4_000_000, n).astype(np.uint32)
    for T in (1.0e5, 8.0e5, 2.0e6, 3.9e6):
        out = np.zeros(n, np.uint8)
        assert L.crt_debug_next_lanes(lanes.ctypes.data, cost.ctypes.data, n, float(T), out.ctypes.data) == 0
        assert np.all(np.isin(out, [64, 32, 16, 8, 4, 2, 1]))
        assert np.all(out[cost > T] <= lanes[cost > T])                    # over the aim: never wider
        assert np.all(out[cost < 0.4 * T] >= lanes[cost < 0.4 * T])         # far below it: never narrower
        for w in (64, 8, 1):                                                # same width, same aim: monotone in the cost
            m = lanes == w
            o = np.argsort(cost[m], kind="stable")
            assert np.all(np.diff(out[m][o].astype(np.int32)) <= 0)
    bad = np.array([3], np.uint8)
    assert L.crt_debug_next_lanes(bad.ctypes.data, cost.ctypes.data, 1, 1.0, np.zeros(1, np.uint8).ctypes.data) != 0


def test_latency_table_solve_fits_the_wavefront_budget(crt):
    """abi.cpp solve_block_table (host logic, no GPU): the aim is the lowest one whose table fits the device's resident wavefronts, never below the most expensive tile's
    one-lane time; a larger budget never gives a higher aim; cheap tiles are not split; with room for everything every expensive tile runs as narrow as the floor asks."""
    L = crt.lib()
    L.crt_debug_solve_block_table.restype = C.c_int
    L.crt_debug_solve_block_table.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_double, C.c_void_p, C.POINTER(C.c_double)]
    rng = np.random.default_rng(5)
    n = 3600
    cost = np.concatenate([rng.integers(50_000, 400_000, n - 300), rng.integers(800_000, 3_400_000, 300)]).astype(np.uint32)   # a sky / floor bulk and 300 tiles on the model
    wide = np.full(n, 64, np.uint8)
    prevT = None
    for budget in (3600.0, 4000.0, 5017.0, 8000.0, 1e9):
        out = np.zeros(n, np.uint8); T = C.c_double(0)
        assert L.crt_debug_solve_block_table(wide.ctypes.data, cost.ctypes.data, n, budget, out.ctypes.data, C.byref(T)) == 0
        waves = int((64 // out.astype(np.int64)).sum())
        assert np.all(np.isin(out, [64, 32, 16, 8, 4, 2, 1]))
        assert waves <= max(budget, n) * 1.001, (budget, waves)
        assert T.value >= 0.48 * cost.max() * 0.999                        # the floor: the most expensive tile as one-lane wavefronts
        assert np.all(out[cost < 0.4 * T.value] == 64)                      # cheap tiles stay one wavefront
        if prevT is not None: assert T.value <= prevT * 1.0001
        prevT = T.value
    assert abs(prevT - 0.48 * cost.max()) <= 0.01 * cost.max()              # unlimited budget: the floor itself
    # measured under a narrowed table: the cost at the width a tile ran is scaled back to its one-wavefront cost before the solve
    lanes = rng.choice(np.array([64, 16, 2], np.uint8), n); out2 = np.zeros(n, np.uint8)
    assert L.crt_debug_solve_block_table(lanes.ctypes.data, cost.ctypes.data, n, 5017.0, out2.ctypes.data, None) == 0
    assert int((64 // out2.astype(np.int64)).sum()) <= 5017 * 1.001
    bad = np.array([5], np.uint8)
    assert L.crt_debug_solve_block_table(bad.ctypes.data, cost.ctypes.data, 1, 10.0, out.ctypes.data, None) != 0


def test_primitive_scene_host_front_matches_the_oracle_and_det_trig_is_accurate(crt, orc):
    """PrimitiveScene's constructor + SetTime on the host front (csrc/host/primitive_scene.cpp) against the oracle's restatement (parity unpinned: both by this repo),
    and the deterministic double-precision acos / cos both the oracle and the HIP kernel use for the torus against libm (<= 1 ulp)."""
    import math
    for t in (0.0, 0.37, 1.3, 12.5):
        ps = crt.HostPrimitiveScene(None); ps.set_time(t)
        o = orc.primitive_scene(None, t)
        assert np.array_equal(ps.state().view(np.uint32), orc.prim_state(o).view(np.uint32)), t
        ps.close()
    acos, cos = orc.det_acos_cos()
    rng = np.random.default_rng(3)
    for x in rng.uniform(-1, 1, 20000):
        assert abs(acos(x) - math.acos(x)) <= np.spacing(math.acos(x))
    for x in rng.uniform(0, 2.35, 20000):
        assert abs(cos(x) - math.cos(x)) <= np.spacing(abs(math.cos(x))) + 1e-17
    assert acos(1.0) == 0.0 and acos(-1.0) == math.pi and cos(0.0) == 1.0
