"""GPU tests against committed golden vectors (incl. outputs of the REAL reference), edge cases, and the C++ facade."""
import ctypes as C
import json
import os
import zlib

import numpy as np
import pytest

from conftest import ASSETS, GOLDEN, scene_path

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["default", "pool_always"])
def render_kernel(request, monkeypatch):
    """every test of this module runs twice: with the back end's default choice of render kernel (render_tiles_kernel for launches of
    <= 64 frames, render_pool_kernel above) and with the stream-pool kernel forced for every launch size"""
    if request.param != "default":
        monkeypatch.setenv("CRT_RENDER_KERNEL", request.param)
    return request.param
G = json.load(open(os.path.join(GOLDEN, "golden.json")))


def crc(a):
    return int(zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff)


def bits(a):
    return np.asarray(a).view(np.uint32)


def write_scene(tmp_path, mesh, name="s.xml", pos=(0.0, -1.0, 2.0), rot=(0.0, 180.0, 0.0), scale=(1.0, 1.0, 1.0), mats=None, extra_objects=()):
    """scene equivalent to tests/golden/make_golden.py::simple_scene but through the XML loader (floor = pavement, sky = gradient)"""
    mats = mats or [(0.0, 0.0, (0.0, 0.0, 0.0), "")]
    objs = [(mesh, 0, pos, rot, scale)] + list(extra_objects)
    o = "".join("<object><model_location>../assets/%s.obj</model_location><material_idx>%d</material_idx><position><x>%r</x><y>%r</y><z>%r</z></position>"
                "<rotation><x>%r</x><y>%r</y><z>%r</z></rotation><scale><x>%r</x><y>%r</y><z>%r</z></scale></object>" % ((m, mi) + tuple(p) + tuple(r) + tuple(s))
                for m, mi, p, r, s in objs)
    ms = "".join("<material><reflectivity>%r</reflectivity><refractivity>%r</refractivity><absorption><x>%r</x><y>%r</y><z>%r</z></absorption><texture_location>%s</texture_location></material>"
                 % ((a, b) + tuple(c) + (t,)) for a, b, c, t in mats)
    p = tmp_path / name
    p.write_text("<scene><scene_name>t</scene_name><light_position><x>0.0</x><y>3.0</y><z>1.0</z></light_position>"
                 "<plane_texture_location>../assets/textures/Stylized_Pavement_basecolor.png</plane_texture_location>"
                 "<skydome_location>../assets/sky_gradient.png</skydome_location><objects>%s</objects><materials>%s</materials></scene>" % (o, ms))
    return str(p)


@pytest.mark.parametrize("mesh", ["bunny", "teapot", "cube"])
def test_gpu_traversal_vs_real_reference_golden(crt, tmp_path, mesh):
    """rays traced by the reference's own BVH::Intersect (oracle/_ref, authoring container) vs the HIP find_nearest"""
    z = np.load(os.path.join(GOLDEN, "ref_bvh_rays.npz"))
    O, D = z[mesh + "_O"], z[mesh + "_D"]
    hs = crt.HostScene(write_scene(tmp_path, mesh), 0, ASSETS)
    b = hs.bvh(0)
    g = G["ref_bvh"][mesh]
    assert (b["nodesUsed"], b["maxDepth"], crc(b["nodes"]), crc(b["triIndices"])) == (g["nodesUsed"], g["maxDepth"], g["nodes"], g["triIndices"])
    ctx = crt.Context(64, 64)
    hs.upload(ctx)
    h = ctx.find_nearest(O, D)
    ro = z[mesh + "_objIdx"]
    both = (h["objIdx"] >= 2) & (ro >= 2)
    assert both.sum() > 200
    for f in ("t", "u", "v"):
        assert np.array_equal(bits(h[f][both]), bits(z[mesh + "_" + f][both])), f
    assert np.array_equal(h["triIdx"][both], z[mesh + "_triIdx"][both])
    miss = (h["objIdx"] == -1) & (ro == -1)
    assert np.array_equal(h["traversed"][miss], z[mesh + "_traversed"][miss]) and np.array_equal(h["tested"][miss], z[mesh + "_tested"][miss])
    assert not ((h["objIdx"] >= 2) & (ro < 2)).any()


@pytest.mark.parametrize("name", sorted(G["orc_render"].keys()))
def test_gpu_render_vs_golden_accumulator(crt, name):
    g = G["orc_render"][name]
    hs = crt.HostScene(scene_path(g["xml"]), g["kind"], ASSETS)
    ctx = crt.Context(g["W"], g["H"], collect_stats=True)
    hs.upload(ctx)
    ctx.render(1, g["frames"], 1)
    acc = ctx.accumulator()
    want = np.load(os.path.join(GOLDEN, "orc_render_%s.npy" % name))
    assert np.abs(acc - want).max() / g["frames"] <= 1e-4
    assert np.array_equal(acc, want)
    assert ctx.counters() == g["counters"]
    px, energy = ctx.resolve_screen(1.0 / (g["frames"] + 1))
    assert crc(px) == g["screen"] and float(np.float32(energy)) == g["energy"]


def test_passes_and_frame_batches_are_order_exact(crt, orc):
    """passes = 3 (three samples per pixel per Tick) and launches of 5 frames (lanes 5..63 idle) against the oracle"""
    hs = crt.HostScene(scene_path("bunny_scene.xml"), 0, ASSETS)
    ctx = crt.Context(64, 48, max_frames_per_launch=5)
    hs.upload(ctx)
    ctx.render(1, 7, 3)                      # spp 1,4,...,19
    acc = ctx.accumulator()
    o, _ = orc.load_scene(scene_path("bunny_scene.xml"), 0, ASSETS)
    o.renderer_init(64, 48)
    o.set_params(5, 3)
    o.render(7, 2)
    assert np.array_equal(acc, o.accumulator())
    assert ctx.timing()["render_launches"] == 2


def test_resolution_not_multiple_of_16_leaves_trailing_pixels_untouched(crt, orc):
    """SCRHEIGHT/16 truncates (renderer.cpp:151): rows >= 16*(H//16) and columns >= 16*(W//16) are never rendered"""
    W, H = 100, 72
    hs = crt.HostScene(scene_path("cube_scene.xml"), 0, ASSETS)
    ctx = crt.Context(W, H)
    hs.upload(ctx)
    ctx.render(1, 2, 1)
    acc = ctx.accumulator()
    assert not acc[64:, :, :].any() and not acc[:, 96:, :].any() and acc[:64, :96, :3].any()
    o, _ = orc.load_scene(scene_path("cube_scene.xml"), 0, ASSETS)
    o.renderer_init(W, H)
    o.render(2, 2)
    assert np.array_equal(acc, o.accumulator())


@pytest.mark.parametrize("depth", [0, 1, 3])
def test_depth_limit(crt, orc, depth):
    hs = crt.HostScene(scene_path("tlas_scene.xml"), 1, ASSETS)
    ctx = crt.Context(64, 48, depth_limit=depth)
    hs.upload(ctx)
    ctx.render(1, 2, 1)
    o, _ = orc.load_scene(scene_path("tlas_scene.xml"), 1, ASSETS)
    o.renderer_init(64, 48)
    o.set_params(depth, 1)
    o.render(2, 2)
    assert np.array_equal(ctx.accumulator(), o.accumulator())
    assert ctx.counters()["rays"] == o.counters()["rays"]


def test_camera_state_and_clear(crt, orc):
    hs = crt.HostScene(scene_path("tlas_scene.xml"), 1, ASSETS)
    ctx = crt.Context(80, 48)
    hs.upload(ctx)
    ctx.render(1, 1, 1)
    ctx.clear()
    ctx.set_camera_state((2.0, 0.5, -1.5), (0.0, -0.5, 3.0))
    ctx.render(1, 2, 1)
    o, _ = orc.load_scene(scene_path("tlas_scene.xml"), 1, ASSETS)
    o.renderer_init(80, 48)
    o.set_camera_state((2.0, 0.5, -1.5), (0.0, -0.5, 3.0))
    o.render(2, 2)
    assert np.array_equal(ctx.accumulator(), o.accumulator())


def test_materials_mirror_dielectric_absorption_texture(crt, orc, tmp_path):
    """every Sample branch: mirror, dielectric with absorption (inside rays, expf), textured diffuse, and a camera inside the glass teapot"""
    xml = write_scene(tmp_path, "teapot", pos=(0.0, -1.0, 2.5), rot=(0.0, 30.0, 0.0), scale=(0.5, 0.5, 0.5),
                      mats=[(0.1, 0.85, (0.8, 0.2, 0.1), ""), (1.0, 0.0, (0.0, 0.0, 0.0), ""), (0.0, 0.0, (0.0, 0.0, 0.0), "../assets/textures/Stylized_Wood_basecolor.tga")],
                      extra_objects=[("cube", 1, (-1.6, -0.5, 3.0), (0.0, 20.0, 0.0), (0.5, 0.5, 0.5)), ("log_fence", 2, (1.8, -1.0, 3.0), (0.0, 0.0, 0.0), (0.6, 0.6, 0.6))])
    for kind in (0, 1):
        hs = crt.HostScene(xml, kind, ASSETS)
        ctx = crt.Context(96, 64, collect_stats=True)
        hs.upload(ctx)
        ctx.render(1, 3, 1)
        o, _ = orc.load_scene(xml, kind, ASSETS)
        o.renderer_init(96, 64)
        o.render(3, 2)
        assert np.array_equal(ctx.accumulator(), o.accumulator()), kind
        assert ctx.counters() == o.counters()


def test_tiny_mesh_root_is_leaf_and_single_blas(crt, orc, tmp_path):
    """2 triangles: BVH root stays a leaf (triCount <= 2), TLAS with one BLAS: root node is a TLAS leaf"""
    obj = tmp_path / "quad.obj"
    obj.write_text("v -1 0 0\nv 1 0 0\nv 1 1.5 0\nv -1 1.5 0\nvn 0 0 -1\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nf 1/1/1 2/2/1 3/3/1 4/4/1\n")
    xml = tmp_path / "quad.xml"
    xml.write_text(open(scene_path("bunny_scene.xml")).read().replace("../assets/bunny.obj", str(obj)).replace("<y>-1.0</y><z>2.0</z>", "<y>-1.0</y><z>3.0</z>"))
    for kind in (0, 1):
        hs = crt.HostScene(str(xml), kind, ASSETS)
        assert hs.bvh(0)["nodesUsed"] == 1
        ctx = crt.Context(64, 48, collect_stats=True)
        hs.upload(ctx)
        ctx.render(1, 2, 1)
        o, _ = orc.load_scene(str(xml), kind, ASSETS)
        o.renderer_init(64, 48)
        o.render(2, 1)
        assert np.array_equal(ctx.accumulator(), o.accumulator()), kind
        assert ctx.counters() == o.counters()
        assert ctx.counters()["mesh_hits"] > 0


def test_many_instances_deep_tlas(crt, orc, tmp_path):
    """40 BLAS instances (cubes / fences / teapots, three materials incl. a mirror and a dielectric) in a ring around the camera: a TLAS
    of 2 x 40 node slots built by agglomerative clustering, rays that enter several overlapping instance boxes, deep unified LDS stack
    (TLAS pushes + return markers + BLAS stack).  Render and FindNearest against the oracle, both scene kinds."""
    import math
    meshes = ["cube", "log_fence", "teapot"]
    extra = []
    for i in range(1, 40):
        a = 2 * math.pi * i / 40
        r = 3.0 + 0.7 * (i % 3)
        extra.append((meshes[i % 3], i % 3, (round(r * math.sin(a), 3), round(-1.0 + 0.35 * (i % 4), 3), round(2.0 + r * math.cos(a), 3)),
                      (0.0, round(37.0 * i % 360, 1), round(5.0 * (i % 5), 1)), (0.3, 0.3 + 0.05 * (i % 3), 0.3)))
    xml = write_scene(tmp_path, "cube", pos=(0.0, -0.5, 5.0), rot=(0.0, 15.0, 0.0), scale=(0.4, 0.4, 0.4),
                      mats=[(0.0, 0.0, (0.0, 0.0, 0.0), ""), (0.9, 0.0, (0.0, 0.0, 0.0), ""), (0.05, 0.9, (0.3, 0.1, 0.5), "")], extra_objects=extra)
    for kind in (1, 0):
        hs = crt.HostScene(xml, kind, ASSETS)
        o, _ = orc.load_scene(xml, kind, ASSETS)
        if kind == 1:
            assert hs.bvh_count() == 40
            (na, ua), (nb, ub) = hs.tlas(), o.tlas()
            assert ua == ub == 80 and np.array_equal(na.view(np.uint8), nb.view(np.uint8))
        ctx = crt.Context(128, 96, collect_stats=True)
        hs.upload(ctx)
        ctx.set_camera_state((0.0, 0.5, 2.0), (0.3, 0.3, 3.0))
        ctx.render(1, 3, 1)
        o.renderer_init(128, 96)
        o.set_camera_state((0.0, 0.5, 2.0), (0.3, 0.3, 3.0))
        o.render(3, 4)
        assert np.array_equal(ctx.accumulator(), o.accumulator()), kind
        assert ctx.counters() == o.counters()
        assert ctx.counters()["mesh_hits"] > 5000
        rng = np.random.default_rng(5)
        O = np.tile(np.array([0.0, 0.5, 2.0], np.float32), (4000, 1)) + rng.uniform(-0.2, 0.2, (4000, 3)).astype(np.float32)
        D = rng.normal(size=(4000, 3)).astype(np.float32); D[:, 1] *= 0.3
        D = (D / np.linalg.norm(D, axis=1, keepdims=True)).astype(np.float32)
        g, c = ctx.find_nearest(O, D), o.find_nearest(O, D)
        for f in ("objIdx", "triIdx", "traversed", "tested"):
            assert np.array_equal(g[f], c[f]), (kind, f)
        for f in ("t", "u", "v"):
            assert np.array_equal(bits(g[f]), bits(c[f])), (kind, f)
        ctx.close()


def test_axis_aligned_rays_nan_exact_slab_path(crt, orc):
    """direction components that are exactly 0 make rD infinite and 0*inf = NaN in the slab test: the kernel must fall back
    to the reference's std::min/std::max operand order (box_exact) and still agree bit for bit"""
    hs = crt.HostScene(scene_path("bunny_scene.xml"), 0, ASSETS)
    ctx = crt.Context(64, 64)
    hs.upload(ctx)
    o, _ = orc.load_scene(scene_path("bunny_scene.xml"), 0, ASSETS)
    b = hs.bvh(0)
    n = b["nodes"][: b["nodesUsed"]]
    pts = np.concatenate([n["aabbMin"], n["aabbMax"]])[:4000].astype(np.float32)      # origins exactly on box planes
    O = np.concatenate([pts - np.array([0, 0, 5], np.float32), pts - np.array([5, 0, 0], np.float32), pts + np.array([0, 5, 0], np.float32)])
    D = np.concatenate([np.tile(np.array([0, 0, 1], np.float32), (len(pts), 1)), np.tile(np.array([1, 0, 0], np.float32), (len(pts), 1)),
                        np.tile(np.array([0, -1, 0], np.float32), (len(pts), 1))])
    g, c = ctx.find_nearest(O, D), o.find_nearest(O, D)
    for f in ("objIdx", "triIdx", "traversed", "tested"):
        assert np.array_equal(g[f], c[f]), f
    for f in ("t", "u", "v"):
        assert np.array_equal(bits(g[f]), bits(c[f])), f
    assert (g["objIdx"] >= 2).sum() > 100


def test_error_codes(crt):
    hs = crt.HostScene(scene_path("cube_scene.xml"), 0, ASSETS)
    ctx = crt.Context(64, 48)
    with pytest.raises(crt.CrtError) as e:
        ctx.render(1, 1, 1)                 # before upload
    assert e.value.code == -5
    hs.upload(ctx)
    with pytest.raises(crt.CrtError) as e:
        ctx.render(1, 1, 9)                 # passes outside 1..4
    assert e.value.code == -1
    ctx.render(1, 0, 1)                     # zero frames: no-op
    assert not ctx.accumulator().any()
    assert len(ctx.find_nearest(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))) == 0
    with pytest.raises(crt.CrtError):
        crt.Context(8, 8)                   # smaller than one tile
    with pytest.raises(crt.CrtError):
        crt.Context(64, 64, tile_first=10, tile_stride=1, tile_count=100)
    with pytest.raises(crt.CrtError):
        crt.Context(64, 64, depth_limit=9)


def test_renderer_facade_tick_semantics(crt, orc):
    """Renderer::Init / Tick / ClearAccumulator through the C++ facade: spp, energy, screen, accumulator as the reference leaves them"""
    hs = crt.HostScene(scene_path("bunny_scene.xml"), 0, ASSETS)
    r = crt.HostRenderer(hs, 96, 64)
    r.init()
    o, _ = orc.load_scene(scene_path("bunny_scene.xml"), 0, ASSETS)
    o.renderer_init(96, 64)
    assert r.spp == 1
    for k in range(3):
        r.tick(16.0)
        o.render(1, 2)
        assert r.spp == o.spp() == k + 2
        assert np.array_equal(r.accumulator(), o.accumulator())
        assert np.array_equal(r.screen(), o.screen())
        assert r.energy == o.energy()
    r.render(4)                              # four Ticks in one submission
    o.render(4, 2)
    assert r.spp == 8 and np.array_equal(r.accumulator(), o.accumulator()) and np.array_equal(r.screen(), o.screen())
    r.clear()
    assert not r.accumulator().any() and r.spp == 8        # ClearAccumulator does not touch spp (renderer.cpp:15-18)
    r.set_camera((0.5, 0.2, -2.5), (0.0, -0.4, 2.0))
    r.tick(16.0)
    o.clear(); o.set_spp(8); o.set_camera_state((0.5, 0.2, -2.5), (0.0, -0.4, 2.0)); o.render(1, 2)
    assert np.array_equal(r.accumulator(), o.accumulator())


def test_multi_window_launch_with_partial_last_window(crt, orc):
    """one crt_render of 150 frames = ONE grid over 3 windows (64 + 64 + 22 frames, a wavefront per (tile, window)); with passes = 2 the
    windows hold 128 samples per pixel.  Both against sequential oracle Ticks, and against window-by-window launches."""
    hs = crt.HostScene(scene_path("bunny_scene.xml"), 0, ASSETS)
    o, _ = orc.load_scene(scene_path("bunny_scene.xml"), 0, ASSETS)
    for passes, frames in [(1, 150), (2, 70)]:
        ctx = crt.Context(64, 48)
        hs.upload(ctx)
        ctx.render(1, frames, passes)
        acc = ctx.accumulator()
        assert ctx.timing()["render_launches"] == 1
        o.renderer_init(64, 48)
        o.set_params(5, passes)
        o.reset_counters()
        o.render(frames, 4)
        assert np.array_equal(acc, o.accumulator()), passes
        assert ctx.counters()["rays"] == o.counters()["rays"]
        c1 = crt.Context(64, 48, max_frames_per_launch=64)
        hs.upload(c1)
        c1.render(1, frames, passes)
        assert np.array_equal(c1.accumulator(), acc) and c1.timing()["render_launches"] == (frames + 63) // 64
        ctx.close(); c1.close()


def test_sample_pool_ring_wraps_with_mixed_launch_sizes(crt, orc):
    """The sample slabs of the launches are regions of ONE ring-allocated pool (8 windows here), recycled FIFO behind GPU-side event waits.
    A burst of asynchronous renders of mixed sizes (1 .. 3 windows per launch, partial windows, several launches per call) wraps the
    ring several times; the image must equal the sequential oracle's bit for bit."""
    hs = crt.HostScene(scene_path("bunny_scene.xml"), 0, ASSETS)
    ctx = crt.Context(64, 48, max_frames_per_launch=192, render_streams=5)
    hs.upload(ctx)
    spp = 1
    for frames in [64, 10, 192, 130, 5, 400, 64, 64, 33, 192, 7]:
        ctx.render(spp, frames, 1)
        spp += frames
    acc = ctx.accumulator()
    assert ctx.timing()["render_launches"] == 1 + 1 + 1 + 1 + 1 + 3 + 1 + 1 + 1 + 1 + 1
    o, _ = orc.load_scene(scene_path("bunny_scene.xml"), 0, ASSETS)
    o.renderer_init(64, 48)
    o.render(spp - 1, 4)
    assert np.array_equal(acc, o.accumulator())
    assert ctx.counters()["rays"] == o.counters()["rays"]


def test_scene_reupload_and_destroy_with_renders_in_flight(crt, orc):
    """crt_upload_scene / crt_destroy while asynchronous launches still read the previous scene's device buffers: both wait for them"""
    a = crt.HostScene(scene_path("bunny_scene.xml"), 0, ASSETS)
    b = crt.HostScene(scene_path("cube_scene.xml"), 0, ASSETS)
    ctx = crt.Context(256, 160)
    a.upload(ctx)
    ctx.render(1, 64, 1)            # in flight ...
    b.upload(ctx)                   # ... while the scene is replaced
    first = ctx.accumulator()
    ctx.clear(); ctx.render(1, 2, 1)
    o, _ = orc.load_scene(scene_path("cube_scene.xml"), 0, ASSETS)
    o.renderer_init(256, 160)
    o.render(2, 4)
    assert np.array_equal(ctx.accumulator(), o.accumulator())
    o2, _ = orc.load_scene(scene_path("bunny_scene.xml"), 0, ASSETS)
    o2.renderer_init(256, 160)
    o2.set_tile_range(37, 1); o2.render(64, 4)
    assert np.array_equal(first[32:48, 80:96], o2.accumulator()[32:48, 80:96])     # tile 37 = (tx 5, ty 2) of the bunny render finished intact
    ctx.render(3, 640, 1)
    ctx.close()                     # destroy with a long launch in flight


@pytest.mark.parametrize("streams", [1, 3, 7])
def test_back_to_back_renders_keep_frame_order(crt, orc, streams):
    """many asynchronous crt_render calls (they overlap on `streams` HIP streams) must accumulate in frame order:
    20 windows of 3 frames, no sync in between, against 60 sequential oracle Ticks"""
    hs = crt.HostScene(scene_path("bunny_scene.xml"), 0, ASSETS)
    ctx = crt.Context(64, 48, render_streams=streams)
    hs.upload(ctx)
    for i in range(20):
        ctx.render(1 + 3 * i, 3, 1)
    acc = ctx.accumulator()
    o, _ = orc.load_scene(scene_path("bunny_scene.xml"), 0, ASSETS)
    o.renderer_init(64, 48)
    o.render(60, 4)
    assert np.array_equal(acc, o.accumulator())
    assert ctx.counters()["rays"] == o.counters()["rays"]
    assert ctx.timing()["render_launches"] == 20
    # a clear between windows is ordered with the accumulates too
    ctx.render(1, 2, 1); ctx.clear(); ctx.render(1, 2, 1)
    o.clear(); o.render(2, 2)
    assert np.array_equal(ctx.accumulator(), o.accumulator())


@pytest.mark.parametrize("xml,kind,W,H", [("cube_scene.xml", 0, 640, 360), ("tlas_scene.xml", 1, 320, 192), ("tlas_scene.xml", 0, 160, 96), ("tower_scene.xml", 0, 200, 120)])
def test_whitted_tick_matches_oracle(crt, orc, xml, kind, W, H):
    """the reference's second front-end (2. WhittedStyle/renderer.cpp) behind the same boundary: deterministic, every pixel of the
    image, reflection / refraction recursion + shadow rays.  640x360 cube = BASELINE config 1."""
    hs = crt.HostScene(scene_path(xml), kind, ASSETS)
    ctx = crt.Context(W, H)
    hs.upload(ctx)
    px = ctx.whitted_tick()
    acc = ctx.accumulator()
    o, _ = orc.load_scene(scene_path(xml), kind, ASSETS)
    o.renderer_init(W, H)
    o.whitted(4)
    want = o.accumulator()
    assert np.isfinite(want).all()
    assert np.abs(acc - want).max() <= 1e-4
    assert np.array_equal(acc, want)
    assert np.array_equal(px, o.screen())
    gc, oc = ctx.counters(), o.counters()
    for k in gc:
        assert gc[k] == oc[k], (k, gc[k], oc[k])
    if xml == "cube_scene.xml" and W == 640:
        assert crc(acc) == G["orc_whitted_cube_640x360"]["acc"] and crc(px) == G["orc_whitted_cube_640x360"]["screen"]
        r = crt.HostRenderer(hs, W, H)                      # same through the C++ Renderer facade
        r.init()
        r.tick_whitted()
        assert np.array_equal(r.accumulator(), want) and np.array_equal(r.screen(), o.screen())
