"""CPU tests that pin the oracle:
  * against the REAL reference compiled in place (oracle/_ref: infra/bvh.cpp, tinyobj, stb_image) — authoring container only;
  * against the committed golden vectors those runs produced (tests/golden/ref_*) — everywhere, including the GPU box;
  * regression vectors for the parts with no executable reference (tests/golden/orc_*, "parity unpinned")."""
import json
import os
import zlib

import numpy as np
import pytest

from conftest import ASSETS, GOLDEN, scene_path

G = json.load(open(os.path.join(GOLDEN, "golden.json")))
MESHES = sorted(G["ref_bvh"].keys())


def crc(a):
    return int(zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff)


def simple_scene(orc, mesh, kind=0):
    o = orc.Oracle(kind)
    o.set_light_position((0, 3, 1))
    o.set_floor_texture(np.full((512, 512), 0x808080, np.uint32))
    o.set_skydome(np.full((4, 8), 0x6080c0, np.uint32))
    o.add_material()
    o.add_object(orc.read_obj(os.path.join(ASSETS, mesh + ".obj")), (0, -1, 2), (0, 180, 0), (1, 1, 1), 0)
    o.build()
    return o


# ---------------------------------------------------------------------------------------------------------------
# golden vectors produced by the real reference
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mesh", MESHES)
def test_obj_reader_matches_tinyobj_golden(orc, mesh):
    pos, nrm, uv = orc.read_obj(os.path.join(ASSETS, mesh + ".obj"))
    g = G["ref_obj"][mesh]
    assert pos.shape[0] == g["corners"]
    assert (crc(pos), crc(nrm), crc(uv)) == (g["pos"], g["nrm"], g["uv"])


@pytest.mark.parametrize("img", sorted(G["ref_img"].keys()))
def test_image_reader_matches_stb_golden(orc, img):
    a = orc.read_image(os.path.join(ASSETS, img))
    assert list(a.shape) == G["ref_img"][img]["shape"]
    assert crc(orc.pack_rgb(a)) == G["ref_img"][img]["packed"]


@pytest.mark.parametrize("mesh", MESHES)
def test_oracle_bvh_build_matches_reference_golden(orc, mesh):
    b = simple_scene(orc, mesh).bvh(0)
    g = G["ref_bvh"][mesh]
    assert crc(b["tris"]) == g["tris_crc"], "triangle soup differs: the build comparison would be meaningless"
    assert (b["nodesUsed"], b["maxDepth"]) == (g["nodesUsed"], g["maxDepth"])
    assert crc(b["nodes"]) == g["nodes"] and crc(b["triIndices"]) == g["triIndices"]


@pytest.mark.parametrize("mesh", ["bunny", "teapot", "cube"])
def test_oracle_traversal_matches_reference_golden(orc, mesh):
    """Rays against BVH::Intersect of the reference.  The oracle's FindNearest tests the light quad and the floor first;
    rays are compared where the mesh is (or is not) the nearest thing in both."""
    z = np.load(os.path.join(GOLDEN, "ref_bvh_rays.npz"))
    O, D = z[mesh + "_O"], z[mesh + "_D"]
    o = simple_scene(orc, mesh)
    h = o.find_nearest(O, D)
    rt, ro = z[mesh + "_t"], z[mesh + "_objIdx"]
    both = (h["objIdx"] >= 2) & (ro >= 2)
    assert both.sum() > 200
    for f in ("t", "u", "v"):
        assert np.array_equal(h[f][both].view(np.uint32), z[mesh + "_" + f][both].view(np.uint32)), f
    assert np.array_equal(h["triIdx"][both], z[mesh + "_triIdx"][both])
    other = (ro >= 2) & ~(h["objIdx"] >= 2)          # reference hit the mesh, oracle reports floor / light: must be nearer
    assert np.all(h["t"][other] < rt[other])
    assert not ((h["objIdx"] >= 2) & (ro < 2)).any()
    miss = (h["objIdx"] == -1) & (ro == -1)             # nothing shortened t before the traversal: loop-trip counters must agree
    assert np.array_equal(h["traversed"][miss], z[mesh + "_traversed"][miss])
    assert np.array_equal(h["tested"][miss], z[mesh + "_tested"][miss])


def camera_oracle(orc, pos_target):
    o = orc.Oracle(0)
    o.renderer_init(1024, 640)                       # the resolution template/camera.h is compiled for (SCRWIDTH x SCRHEIGHT)
    if pos_target is not None:
        o.set_camera_state(*pos_target)
    return o


@pytest.mark.parametrize("name", sorted(G["ref_camera"]["cams"].keys()))
def test_oracle_camera_matches_reference_golden(orc, name):
    """Camera(): default frustum, SetCameraState and GetPrimaryRay of the real reference (template/camera.h:14-30, 61-73)"""
    g = G["ref_camera"]["cams"][name]
    xy = np.load(os.path.join(GOLDEN, "ref_camera_xy.npy"))
    assert crc(xy) == G["ref_camera"]["xy"]
    o = camera_oracle(orc, g["pos_target"])
    assert crc(np.stack(o.camera())) == g["corners"]
    O, D = o.primary_rays(xy)
    assert (crc(O), crc(D)) == (g["O"], g["D"])


def test_oracle_texture_matches_reference_golden(orc):
    """Texture::LoadFromFile's packing, Texture::Sample and Material::GetAlbedo of the real reference (template/texture.h, material.h)"""
    g = G["ref_texture"]
    tex = orc.pack_rgb(orc.read_image(os.path.join(ASSETS, g["file"])))
    assert crc(tex) == g["texels"]
    assert crc(orc.pack_rgb(orc.read_image(os.path.join(ASSETS, "textures/Stylized_Wood_basecolor.tga")))) == g["tga"]
    uv = np.load(os.path.join(GOLDEN, "ref_texture_uv.npy"))
    assert crc(uv) == g["uv"]
    rgb = orc.texture_sample(tex, uv)
    assert crc(rgb) == g["rgb"] == g["albedo"]


def deform(p):
    """vertex animation stand-in (same as tests/golden/make_golden.py): p + 0.1 * (p.yzx * p.zxy), float32 products and sums only"""
    p = np.asarray(p, np.float32)
    return (p + np.float32(0.1) * (p[..., [1, 2, 0]] * p[..., [2, 0, 1]])).astype(np.float32)


@pytest.mark.parametrize("mesh", sorted(G["ref_refit"].keys()))
def test_oracle_refit_matches_reference_golden(orc, mesh):
    """BVH::Refit (infra/bvh.cpp:26-43) of the real reference after moving the vertices, incl. its skipped node 1"""
    o = simple_scene(orc, mesh)
    t = o.bvh(0)["tris"]
    moved = deform(np.stack([t["vertex0"], t["vertex1"], t["vertex2"]], axis=1))
    assert crc(moved) == G["ref_refit"][mesh]["moved"]
    o.move_and_refit(0, moved)
    b = o.bvh(0)
    assert crc(b["nodes"]) == G["ref_refit"][mesh]["nodes"]
    assert np.array_equal(b["tris"]["vertex1"], moved[:, 1])


# ---------------------------------------------------------------------------------------------------------------
# live comparison with the real reference (authoring container)
# ---------------------------------------------------------------------------------------------------------------
def test_oracle_camera_and_texture_vs_reference_live(orc, ref):
    rng = np.random.default_rng(3)
    xy = rng.uniform(-2, [1026, 642], (4000, 2)).astype(np.float32)
    for pt in (None, ((0.5, 1.25, -3.0), (0.0, 0.0, 2.0)), ((-2.0, 3.0, 1.0), (0.0, -1.0, 2.5))):
        corners, O, D = ref.camera_rays(xy, pt)
        o = camera_oracle(orc, pt)
        assert np.array_equal(np.stack(o.camera()).view(np.uint32), corners.view(np.uint32))
        oO, oD = o.primary_rays(xy)
        assert np.array_equal(oO.view(np.uint32), O.view(np.uint32)) and np.array_equal(oD.view(np.uint32), D.view(np.uint32))
    for f in ("textures/Stylized_Pavement_basecolor.png", "textures/Stylized_Wood_basecolor.tga", "textures/Defuse_wok.png"):
        tex = ref.texture_load(os.path.join(ASSETS, f))
        assert np.array_equal(tex, orc.pack_rgb(orc.read_image(os.path.join(ASSETS, f)))), f
    uv = rng.uniform(-0.5, 1.5, (20000, 2)).astype(np.float32)
    small = rng.integers(0, 1 << 24, (37, 53)).astype(np.uint32)
    for tex in (ref.texture_load(os.path.join(ASSETS, "textures/Stylized_Pavement_basecolor.png")), small):
        rgb, alb = ref.texture_sample(tex, uv)
        assert np.array_equal(orc.texture_sample(tex, uv).view(np.uint32), rgb.view(np.uint32)) and np.array_equal(rgb, alb)


@pytest.mark.parametrize("mesh", ["bunny", "wok"])
def test_oracle_refit_vs_reference_live(orc, ref, mesh):
    o = simple_scene(orc, mesh)
    t = o.bvh(0)["tris"]
    h, _ = ref.bvh_build(t)
    try:
        moved = deform(np.stack([t["vertex0"], t["vertex1"], t["vertex2"]], axis=1))
        rn = ref.bvh_move_and_refit(h, moved)
        o.move_and_refit(0, moved)
        assert np.array_equal(rn.view(np.uint8), o.bvh(0)["nodes"].view(np.uint8))
    finally:
        ref.bvh_free(h)



@pytest.mark.parametrize("mesh", ["bunny", "wok", "teapot"])
def test_oracle_vs_reference_live(orc, ref, mesh):
    o = simple_scene(orc, mesh)
    b = o.bvh(0)
    h, rb = ref.bvh_build(b["tris"])
    try:
        assert rb["nodesUsed"] == b["nodesUsed"] and rb["maxDepth"] == b["maxDepth"]
        assert np.array_equal(rb["nodes"].view(np.uint8), b["nodes"].view(np.uint8))
        assert np.array_equal(rb["triIndices"], b["triIndices"])
        rng = np.random.default_rng(99)
        n = 20000
        O = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
        O[:, 1] = np.abs(O[:, 1]) + 0.2
        T = rng.uniform(-1, 1, (n, 3)).astype(np.float32) + np.array([0, -0.3, 2], np.float32)
        D = T - O
        D = (D / np.linalg.norm(D, axis=1, keepdims=True)).astype(np.float32)
        rh = ref.bvh_intersect(h, O, D)
        oh = o.find_nearest(O, D)
        both = (oh["objIdx"] >= 2) & (rh["objIdx"] >= 2)
        assert both.sum() > 1000
        for f in ("t", "u", "v"):
            assert np.array_equal(oh[f][both].view(np.uint32), rh[f][both].view(np.uint32))
        assert np.array_equal(oh["triIdx"][both], rh["triIdx"][both])
    finally:
        ref.bvh_free(h)


def test_python_readers_vs_reference_live(orc, ref):
    for m in MESHES:
        mine, theirs = orc.read_obj(os.path.join(ASSETS, m + ".obj")), ref.obj_load(os.path.join(ASSETS, m + ".obj"))
        for a, b in zip(mine, theirs):
            assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), m
    for f in G["ref_img"]:
        assert np.array_equal(orc.read_image(os.path.join(ASSETS, f)), ref.image_load(os.path.join(ASSETS, f))), f


# ---------------------------------------------------------------------------------------------------------------
# oracle regression vectors (integrator etc.: no executable reference exists — "parity unpinned")
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(G["orc_render"].keys()))
def test_oracle_render_regression(orc, name):
    g = G["orc_render"][name]
    o, _ = orc.load_scene(scene_path(g["xml"]), g["kind"], ASSETS)
    o.renderer_init(g["W"], g["H"])
    o.render(g["frames"], 2)
    acc = o.accumulator()
    assert np.array_equal(acc, np.load(os.path.join(GOLDEN, "orc_render_%s.npy" % name)))
    assert crc(acc) == g["acc"] and crc(o.screen()) == g["screen"]
    assert o.counters() == g["counters"]
    assert float(np.float32(o.energy())) == g["energy"]
    seeds = [o.tile_seed(g["frames"], t) for t in range((g["W"] // 16) * (g["H"] // 16))]
    assert crc(np.array(seeds, np.uint32)) == g["last_frame_tile_seeds"]


def test_oracle_thread_count_invariance(orc):
    """renderer.cpp:120 seeds one stream per (tile, frame): the image cannot depend on the tile schedule."""
    res = []
    for threads in (1, 5):
        o, _ = orc.load_scene(scene_path("bunny_scene.xml"), 0, ASSETS)
        o.renderer_init(80, 64)
        o.render(3, threads)
        res.append(o.accumulator())
    assert np.array_equal(res[0], res[1])


def test_whitted_config1_regression(orc):
    """BASELINE config 1: cube.obj FileScene, Whitted-style, 640x360, CPU reference path (plumbing, no GPU)."""
    o, _ = orc.load_scene(scene_path("cube_scene.xml"), 0, ASSETS)
    o.renderer_init(640, 360)
    o.whitted(4)
    g = G["orc_whitted_cube_640x360"]
    acc = o.accumulator()
    assert np.isfinite(acc).all() and acc[..., :3].max() > 0.5
    assert crc(acc) == g["acc"] and crc(o.screen()) == g["screen"] and o.counters() == g["counters"]


# ---------------------------------------------------------------------------------------------------------------
# small known-answer checks of the restated primitives
# ---------------------------------------------------------------------------------------------------------------
def test_rng_known_answers(orc):
    L = orc.lib()

    def wang(s):
        s = ((s ^ 61) ^ (s >> 16)) & 0xffffffff
        s = (s * 9) & 0xffffffff
        s ^= s >> 4
        s = (s * 0x27d4eb2d) & 0xffffffff
        s ^= s >> 15
        return s
    for base in (0, 1, 1799, 123456789, 0xffffffff):
        assert L.orc_init_seed(base) == wang(((base + 1) * 17) & 0xffffffff)
    import ctypes as C
    s = C.c_uint32(0x12345678)
    x = 0x12345678
    for _ in range(100):
        x ^= (x << 13) & 0xffffffff
        x ^= x >> 17
        x ^= (x << 5) & 0xffffffff
        assert L.orc_random_uint(C.byref(s)) == x and s.value == x


def test_deterministic_math_accuracy(orc):
    """crt_expf / crt_atan2f / crt_acosf are the path's only transcendental functions (absorption, skydome lookup);
    they must stay within a few ulp of the correctly rounded result."""
    L = orc.lib()
    rng = np.random.default_rng(5)

    def ulps(got, want):
        got = np.asarray(got, np.float32)
        want32 = want.astype(np.float32)
        sp = np.spacing(np.abs(want32)).astype(np.float64)
        return np.abs(got.astype(np.float64) - want) / np.maximum(sp, 1e-45)
    x = np.concatenate([rng.uniform(-40, 0, 4000), rng.uniform(-90, 80, 2000), [0.0, -0.0, -1e-8]]).astype(np.float32)
    e = np.array([L.orc_expf(float(v)) for v in x], np.float32)
    assert ulps(e, np.exp(x.astype(np.float64))).max() <= 2.0
    assert L.orc_expf(0.0) == 1.0 and L.orc_expf(-0.0) == 1.0
    y = rng.uniform(-1, 1, 6000).astype(np.float32)
    xx = rng.uniform(-1, 1, 6000).astype(np.float32)
    a = np.array([L.orc_atan2f(float(p), float(q)) for p, q in zip(y, xx)], np.float32)
    assert np.abs(a.astype(np.float64) - np.arctan2(y.astype(np.float64), xx.astype(np.float64))).max() <= 6e-7
    c = np.concatenate([rng.uniform(-1, 1, 6000), [1.0, -1.0, 0.0, 0.5, -0.5]]).astype(np.float32)
    ac = np.array([L.orc_acosf(float(v)) for v in c], np.float32)
    assert np.abs(ac.astype(np.float64) - np.arccos(c.astype(np.float64))).max() <= 6e-7
    assert L.orc_atan2f(0.0, -1.0) == np.float32(np.pi) and L.orc_atan2f(0.0, 1.0) == 0.0


def test_reference_struct_sizes(orc):
    assert orc.TRI_DTYPE.itemsize == 112 and orc.NODE_DTYPE.itemsize == 32 and orc.TLAS_DTYPE.itemsize == 32


# ---------------------------------------------------------------------------------------------------------------
# template/tmplmath.h inline functions on the path + infra/helper.h's Vertex table: pinned to the
# real reference through oracle/_ref (ref_math_probe / ref_vertex_dedup) and its committed outputs tests/golden/ref_math.npz, ref_vertex.npz
# ---------------------------------------------------------------------------------------------------------------
MATH_FIELDS = [("normalize", 0, 3), ("reflect", 3, 6), ("cross", 6, 9), ("dot", 9, 10), ("mat4::Translate", 10, 26), ("mat4::RotateX", 26, 42),
               ("mat4::RotateY", 42, 58), ("mat4::RotateZ", 58, 74), ("mat4::Scale", 74, 90), ("FastInvertedTransformNoScale", 90, 106),
               ("aabb::Grow(float3) / Area", 106, 113), ("aabb::Grow(aabb) / Area", 113, 120)]


def _same_bits(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.uint32); b = np.ascontiguousarray(b, np.float32).view(np.uint32)
    nan = lambda x: (x & 0x7fffffff) > 0x7f800000                      # any NaN matches any NaN (payload / sign of an invalid operation are not specified)
    return bool(np.all((a == b) | (nan(a) & nan(b))))


@pytest.mark.parametrize("who", ["oracle", "host_front"])
def test_tmplmath_inlines_match_reference_golden(orc, who):
    z = np.load(os.path.join(GOLDEN, "ref_math.npz"))
    assert crc(z["outputs"]) == G["ref_math"]["outputs"] and len(z["inputs"]) == G["ref_math"]["rows"]
    if who == "oracle":
        got = orc.math_probe(z["inputs"])
    else:
        from conftest import load_crt
        got = load_crt().host_math_probe(z["inputs"])
    for name, a, b in MATH_FIELDS:
        assert _same_bits(got[:, a:b], z["outputs"][:, a:b]), name


@pytest.mark.parametrize("who", ["oracle", "host_front"])
def test_vertex_table_matches_reference_golden(orc, who):
    """Vertex::operator== is float equality (+0 == -0: the first occurrence's bits are kept; a NaN vertex equals nothing and model.cpp:50's second
    lookup then yields index 0): corner indices and the unique-vertex array of the real std::unordered_map<Vertex, uint32_t>"""
    z = np.load(os.path.join(GOLDEN, "ref_vertex.npz"))
    assert crc(z["idx"]) == G["ref_math"]["vertex_idx"] and len(z["unique"]) == G["ref_math"]["vertex_unique"]
    if who == "oracle":
        idx, uniq = orc.vertex_dedup(z["corners"])
    else:
        from conftest import load_crt
        idx, uniq = load_crt().host_vertex_dedup(z["corners"])
    assert np.array_equal(idx, z["idx"])
    assert uniq.shape == z["unique"].shape and _same_bits(uniq, z["unique"])


def test_tmplmath_and_vertex_live(orc):
    """authoring container only: the same probes against oracle/_ref directly, on fresh random inputs"""
    try:
        ref = orc.Ref()
    except FileNotFoundError:
        pytest.skip("oracle/_ref is built only where /root/reference is mounted")
    rng = np.random.default_rng(5)
    x = rng.uniform(-10, 10, (3000, 12)).astype(np.float32)
    assert _same_bits(orc.math_probe(x), ref.math_probe(x))
    v = np.round(rng.uniform(-1, 1, (4000, 8)) * 4).astype(np.float32) / 4          # many exact duplicates, +0 and -0
    i1, u1 = orc.vertex_dedup(v)
    i2, u2, _ = ref.vertex_dedup(v)
    assert np.array_equal(i1, i2) and _same_bits(u1, u2)


def test_sky_lookup_deviation_from_libm_is_bounded(orc):
    """GetSkyColor (file_scene.cpp:142-154) calls libm's atan2 / acos, the path here its own deterministic polynomials (so that x86 and gfx950
    agree bit for bit).  The two differ in the last ulp of phi / theta, which moves the nearest-texel index (int)(u * w) only when u * w lands within
    that ulp of an integer.  This test measures how often, on a 4096 x 2048 skydome: DESIGN.md quotes the bound."""
    L = orc.lib()
    rng = np.random.default_rng(11)
    n = 200000
    d = rng.normal(size=(n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    PI, INV2PI, INVPI = np.float32(3.14159265358979323846264), np.float32(0.15915494309189533576888), np.float32(0.31830988618379067153777)
    phi_det = np.array([L.orc_atan2f(float(-z), float(x)) for x, z in zip(d[:, 0], d[:, 2])], np.float32) + PI
    th_det = np.array([L.orc_acosf(float(-y)) for y in d[:, 1]], np.float32)
    phi_lm = np.arctan2(-d[:, 2].astype(np.float64), d[:, 0].astype(np.float64)).astype(np.float32) + PI      # correctly rounded = what a good libm returns
    th_lm = np.arccos(-d[:, 1].astype(np.float64)).astype(np.float32)
    W, H = 4096, 2048

    def texel(phi, th):
        u = np.clip(phi * INV2PI, 0, 1); v = np.float32(1) - np.clip(th * INVPI, 0, 1)
        return np.clip((u * np.float32(W)).astype(np.int32), 0, W - 1), np.clip((v * np.float32(H)).astype(np.int32), 0, H - 1)
    xd, yd = texel(phi_det, th_det); xl, yl = texel(phi_lm, th_lm)
    flips = int(((xd != xl) | (yd != yl)).sum())
    assert np.abs(xd - xl).max() <= 1 and np.abs(yd - yl).max() <= 1          # never more than the neighbouring texel
    assert flips <= n * 5e-4, flips                                           # measured: 11 of 200 000 directions (1 in 18 000) at this resolution
    assert np.abs(phi_det.astype(np.float64) - phi_lm).max() <= 6e-7 and np.abs(th_det.astype(np.float64) - th_lm).max() <= 6e-7   # <= 2.5 ulp at pi
    print("sky texel flips vs libm: %d of %d (%.4f %%)" % (flips, n, 100.0 * flips / n))


# ---------------------------------------------------------------------------------------------------------------
# FileScene's alternative accelerators: KDTree (infra/kdtree.cpp) and Grid (infra/grid.cpp), pinned to the real reference
# (both files compile unmodified in oracle/_ref): structure CRCs in golden.json "ref_alt", the reference's own hits in ref_alt_rays.npz
# ---------------------------------------------------------------------------------------------------------------
def _alt_matches(d, g, kind):
    if kind == "kd":
        return (crc(d["nodes"]), crc(d["refs"]), len(d["nodes"]), len(d["refs"]), d["maxDepth"], d["nodesUsed"]) == (g["nodes"], g["refs"], g["nodeCount"], g["refCount"], g["maxDepth"], g["nodesUsed"])
    return ([int(x) for x in d["resolution"]], crc(d["cellSize"]), crc(d["boundsMin"]), crc(d["boundsMax"]), crc(d["cellStart"]), crc(d["refs"])) == \
           (g["resolution"], g["cellSize"], g["boundsMin"], g["boundsMax"], g["cellStart"], g["refs"])


@pytest.mark.parametrize("kind", ["kd", "grid"])
@pytest.mark.parametrize("mesh", ["bunny", "teapot", "cube"])
def test_oracle_alt_accel_matches_reference_golden(orc, mesh, kind):
    tris = simple_scene(orc, mesh).bvh(0)["tris"]
    a = orc.alt_accel(kind, tris)
    assert _alt_matches(a.dump(), G["ref_alt"][mesh][kind], kind)
    z = np.load(os.path.join(GOLDEN, "ref_bvh_rays.npz")); r = np.load(os.path.join(GOLDEN, "ref_alt_rays.npz"))
    h = a.intersect(z[mesh + "_O"], z[mesh + "_D"])
    a.close()
    for f in ("t", "u", "v"):
        assert np.array_equal(h[f].view(np.uint32), r["%s_%s_%s" % (mesh, kind, f)].view(np.uint32)), f
    for f in ("objIdx", "triIdx", "traversed", "tested"):
        assert np.array_equal(h[f], r["%s_%s_%s" % (mesh, kind, f)]), f


@pytest.mark.parametrize("kind,code", [("kd", 1), ("grid", 2)])
@pytest.mark.parametrize("mesh", ["bunny", "cube"])
def test_host_front_alt_accel_build_matches_reference_golden(tmp_path, mesh, kind, code):
    from conftest import load_crt
    from test_gpu_golden_and_edges import write_scene as write_simple_scene
    crt = load_crt()
    hs = crt.HostScene(write_simple_scene(tmp_path, mesh), 0, ASSETS)
    assert _alt_matches(hs.build_alt(code), G["ref_alt"][mesh][kind], kind)


def test_alt_accel_live(orc):
    """authoring container only: oracle vs the real kdtree.cpp / grid.cpp on a mesh and rays the goldens do not cover"""
    try:
        ref = orc.Ref()
    except FileNotFoundError:
        pytest.skip("oracle/_ref is built only where /root/reference is mounted")
    tris = simple_scene(orc, "log_fence").bvh(0)["tris"]
    rng = np.random.default_rng(9)
    O = rng.uniform(-3, 3, (2000, 3)).astype(np.float32); O[:, 1] = np.abs(O[:, 1]) + 0.1
    D = (np.array([0, -0.5, 2], np.float32) + rng.uniform(-1, 1, (2000, 3)).astype(np.float32)) - O
    D = (D / np.linalg.norm(D, axis=1, keepdims=True)).astype(np.float32)
    D[:20, 0] = 0; D[20:40, 1] = 0; D[40:60, 2] = 0                   # axis-parallel components: infinite reciprocal, NaN plane distances
    for kind in ("kd", "grid"):
        a, b = ref.alt_accel(kind, tris), orc.alt_accel(kind, tris)
        da, db = a.dump(), b.dump()
        assert all(np.array_equal(da[k], db[k]) if isinstance(da[k], np.ndarray) else da[k] == db[k] for k in da)
        ha, hb = a.intersect(O, D), b.intersect(O, D)
        assert all(np.array_equal(ha[f].view(np.uint32), hb[f].view(np.uint32)) for f in ha.dtype.names), kind
        a.close(); b.close()
