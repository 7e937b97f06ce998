"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Bar: integer / index results bit-exact; the float accumulator bit-exact as well (the design is bit-reproducible:
no FMA contraction, IEEE div/sqrt, ordered accumulation) — the 1e-4 per-channel gate of BASELINE.json's north_star
is asserted too, as the contractual tolerance."""
import numpy as np
import pytest

from conftest import ASSETS, scene_path

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["default", "pool_always"])
def render_kernel(request, monkeypatch):
    """every test of this module runs twice: with the back end's default choice of render kernel (render_tiles_kernel for launches of
    <= 64 frames, render_pool_kernel above) and with the stream-pool kernel forced for every launch size"""
    if request.param != "default":
        monkeypatch.setenv("CRT_RENDER_KERNEL", request.param)
    return request.param

TOL = 1e-4   # north_star: "output matches the CPU reference at a fixed RNG seed within 1e-4 per channel"


def _rays(n, seed, target=(0.0, -0.3, 2.5), spread=1.5):
    rng = np.random.default_rng(seed)
    O = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    O[:, 1] = np.abs(O[:, 1]) * 0.8 - 0.5
    T = rng.uniform(-spread, spread, (n, 3)).astype(np.float32) + np.asarray(target, np.float32)
    D = T - O
    D = (D / np.linalg.norm(D, axis=1, keepdims=True)).astype(np.float32)
    return O, D


def _same_bits(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


def _eq_pm0(a, b):
    """bit-equal up to the sign of zero"""
    a = np.asarray(a)
    b = np.asarray(b)
    return np.array_equal(a, b) and not np.isnan(a).any()


@pytest.mark.parametrize("xml,kind", [("bunny_scene.xml", 0), ("cube_scene.xml", 0), ("tlas_scene.xml", 1), ("tlas_scene.xml", 0), ("tower_scene.xml", 0)])
def test_find_nearest_bit_exact(crt, orc, xml, kind):
    hs = crt.HostScene(scene_path(xml), kind, ASSETS)
    ctx = crt.Context(64, 64)
    hs.upload(ctx)
    o, _ = orc.load_scene(scene_path(xml), kind, ASSETS)
    O, D = _rays(50000, 7)
    g = ctx.find_nearest(O, D)
    c = o.find_nearest(O, D)
    for f in ("objIdx", "triIdx", "traversed", "tested"):
        assert np.array_equal(g[f], c[f]), f
    for f in ("t", "u", "v"):
        assert _same_bits(g[f], c[f]), f
    assert (g["objIdx"] >= 2).sum() > 1000      # the test actually exercises mesh hits
    gc, oc = ctx.counters(), o.counters()
    for k in ("rays", "interior_iters", "leaf_iters", "tri_tests", "tlas_iters", "blas_visits", "mesh_hits"):
        assert gc[k] == oc[k], k


@pytest.mark.parametrize("xml,kind,W,H,frames", [("bunny_scene.xml", 0, 128, 96, 3), ("tlas_scene.xml", 1, 128, 96, 2), ("cube_scene.xml", 0, 80, 48, 5),
                                                   ("tower_scene.xml", 0, 128, 96, 2)])
def test_render_matches_oracle(crt, orc, xml, kind, W, H, frames):
    hs = crt.HostScene(scene_path(xml), kind, ASSETS)
    ctx = crt.Context(W, H, collect_stats=True)
    hs.upload(ctx)
    ctx.render(1, frames, 1)
    acc = ctx.accumulator()
    o, _ = orc.load_scene(scene_path(xml), kind, ASSETS)
    o.renderer_init(W, H)
    o.render(frames, threads=4)
    ref = o.accumulator()
    assert np.isfinite(ref).all()
    err = np.abs(acc - ref).max() / frames
    assert err <= TOL, "max per-sample-normalised channel error %g" % err
    assert _eq_pm0(acc, ref), "accumulator not bit-identical (max abs diff %g)" % np.abs(acc - ref).max()
    gc, oc = ctx.counters(), o.counters()
    for k in gc:
        assert gc[k] == oc[k], (k, gc[k], oc[k])
    # screen + energy as ProcessTile / Tick leave them
    px, energy = ctx.resolve_screen(1.0 / (frames + 1))
    assert np.array_equal(px, o.screen())
    assert energy == o.energy()


@pytest.mark.parametrize("switch", ["CRT_DEBUG_GENERAL_PRIMS", "CRT_DEBUG_NO_ROOTPAIR"])
@pytest.mark.parametrize("xml,kind", [("bunny_scene.xml", 0), ("tlas_scene.xml", 1), ("light_at_origin", 0)])
def test_general_code_paths_are_bit_identical(crt, monkeypatch, tmp_path, switch, xml, kind):
    """The render kernel takes two shortcuts that crt_upload_scene enables per scene: the short quad / plane tests when the light
    is an unrotated quad and the floor normal is (0,1,0) (what FileScene / TLASFileScene always build), and every ray's first
    traversal step from the root's child pair held in the kernel arguments.  The debug switches force the general expressions /
    the plain root start on the same scenes: image, ray count and traversal counters must not change by a bit."""
    W, H, frames = 128, 96, 3
    if xml == "light_at_origin":      # light quad in the plane y = 0 through the origin: the translation terms of its invT are zeros (of either sign)
        p = tmp_path / "l0.xml"
        p.write_text(open(scene_path("bunny_scene.xml")).read().replace("<light_position><x>0.0</x><y>3.0</y><z>1.0</z></light_position>",
                                                                         "<light_position><x>0.0</x><y>0.0</y><z>0.0</z></light_position>"))
        assert "<y>0.0</y><z>0.0</z></light_position>" in p.read_text()
        hs = crt.HostScene(str(p), kind, ASSETS)
    else:
        hs = crt.HostScene(scene_path(xml), kind, ASSETS)
    out = []
    for forced in (False, True):
        if forced:
            monkeypatch.setenv(switch, "1")
        ctx = crt.Context(W, H, collect_stats=True)
        hs.upload(ctx)                       # the switches are read by crt_upload_scene
        ctx.render(1, frames, 1)
        out.append((ctx.accumulator(), ctx.counters()))
        ctx.close()
    monkeypatch.delenv(switch)
    assert np.array_equal(out[0][0], out[1][0])
    assert out[0][1] == out[1][1]


@pytest.mark.parametrize("xml,kind", [("bunny_scene.xml", 0), ("tlas_scene.xml", 1)])
def test_render_after_refit_matches_oracle(crt, orc, xml, kind):
    """moved vertices -> Refit on the CPU (host front) -> crt_host_scene_upload again -> the HIP path renders the refitted scene
    exactly as the oracle does (the oracle's Refit is pinned to the real reference's, tests/golden ref_refit)"""
    from test_oracle_pinning import deform
    W, H, frames = 128, 96, 2
    hs = crt.HostScene(scene_path(xml), kind, ASSETS)
    o, _ = orc.load_scene(scene_path(xml), kind, ASSETS)
    ctx = crt.Context(W, H, collect_stats=True)
    hs.upload(ctx)
    ctx.render(1, frames, 1)
    before = ctx.accumulator()
    i = hs.bvh_count() - 1
    t = hs.bvh(i)["tris"]
    moved = deform(np.stack([t["vertex0"], t["vertex1"], t["vertex2"]], axis=1))
    hs.move_and_refit(i, moved)
    o.move_and_refit(i, moved)
    hs.upload(ctx)
    ctx.clear(); ctx.reset_counters()
    ctx.render(1, frames, 1)
    acc = ctx.accumulator()
    o.renderer_init(W, H)
    o.render(frames, threads=4)
    assert not np.array_equal(acc, before)
    assert _eq_pm0(acc, o.accumulator())
    gc, oc = ctx.counters(), o.counters()
    for k in gc:
        assert gc[k] == oc[k], (k, gc[k], oc[k])
