"""animation behind the same boundary.  Instance motion = BLASBVH::SetTransform (infra/blas_bvh.cpp:363-374) + TLASBVH::Build per frame
(infra/tlas_bvh.cpp:17-55) on the host, then crt_update_scene(CRT_UPDATE_TRANSFORMS) rewrites the TLAS / instance sections of the device scene in place;
moved vertices = Refit (infra/bvh.cpp:26-43) + crt_update_scene(CRT_UPDATE_BOUNDS).  Every frame is checked against the oracle doing the same."""
import time

import numpy as np
import pytest

from conftest import ASSETS, scene_path

pytestmark = pytest.mark.gpu


def rigid(angle_y, t):
    c, s = np.float32(np.cos(np.float32(angle_y))), np.float32(np.sin(np.float32(angle_y)))
    m = np.eye(4, dtype=np.float32)
    m[0, 0] = c; m[0, 2] = s; m[2, 0] = -s; m[2, 2] = c; m[:3, 3] = t
    return m


@pytest.fixture(params=["default", "pool_always"])
def kernel(request, monkeypatch):
    if request.param != "default":
        monkeypatch.setenv("CRT_RENDER_KERNEL", request.param)
    return request.param


def test_instance_motion_three_frames_matches_oracle(crt, orc, kernel):
    """an `animating` loop (renderer.cpp:147): per frame one instance gets a new rigid transform, the TLAS is rebuilt, the accumulator is cleared and
    one Tick rendered — without re-uploading the scene"""
    W, H = 128, 96
    hs = crt.HostScene(scene_path("tlas_scene.xml"), 1, ASSETS)
    o, _ = orc.load_scene(scene_path("tlas_scene.xml"), 1, ASSETS)
    o.renderer_init(W, H)
    ctx = crt.Context(W, H, collect_stats=True)
    hs.upload(ctx)
    T0 = hs.blas_transform(1)[0].reshape(4, 4)
    prev = None
    for f in range(3):
        T = rigid(0.4 * (f + 1), T0[:3, 3] + np.array([0.3 * f, 0.1 * f, -0.2 * f], np.float32))
        hs.set_transform(1, T); o.set_transform(1, T)
        assert np.array_equal(hs.tlas()[0], o.tlas()[0])                   # same TLASBVH::Build on both sides
        hs.update(ctx, crt.UPDATE_TRANSFORMS)
        ctx.clear(); ctx.reset_counters(); o.clear(); o.reset_counters()
        ctx.render(1, 2, 1); o.render(2, 4)
        acc = ctx.accumulator()
        assert np.array_equal(acc, o.accumulator()), f
        assert ctx.counters() == o.counters()
        assert prev is None or not np.array_equal(acc, prev)
        prev = acc
    # the query entry sees the moved instance too
    rng = np.random.default_rng(3)
    O = rng.uniform(-3, 3, (500, 3)).astype(np.float32); O[:, 1] = np.abs(O[:, 1]) + 0.2
    D = (np.array([0, -0.3, 2], np.float32) + rng.uniform(-1, 1, (500, 3)).astype(np.float32)) - O
    D = (D / np.linalg.norm(D, axis=1, keepdims=True)).astype(np.float32)
    g, w = ctx.find_nearest(O, D), o.find_nearest(O, D)
    for k in ("t", "u", "v", "objIdx", "triIdx"):
        assert np.array_equal(g[k], w[k]), k


@pytest.mark.parametrize("xml,kind", [("bunny_scene.xml", 0), ("tlas_scene.xml", 1)])
def test_refit_in_place_matches_oracle_and_reupload(crt, orc, kernel, xml, kind):
    from test_oracle_pinning import deform
    W, H = 96, 64
    hs = crt.HostScene(scene_path(xml), kind, ASSETS)
    o, _ = orc.load_scene(scene_path(xml), kind, ASSETS)
    o.renderer_init(W, H)
    ctx = crt.Context(W, H)
    hs.upload(ctx)
    ctx.render(1, 2, 1)
    i = hs.bvh_count() - 1
    t = hs.bvh(i)["tris"]
    moved = deform(np.stack([t["vertex0"], t["vertex1"], t["vertex2"]], axis=1))
    hs.move_and_refit(i, moved); o.move_and_refit(i, moved)
    hs.update(ctx, crt.UPDATE_BOUNDS)                                                       # in place, renders of the old scene may still be in flight
    ctx.clear()
    ctx.render(1, 3, 1); o.render(3, 4)
    acc = ctx.accumulator()
    assert np.array_equal(acc, o.accumulator())
    c2 = crt.Context(W, H); hs.upload(c2); c2.render(1, 3, 1)                                  # a fresh upload of the refitted scene gives the same image
    assert np.array_equal(acc, c2.accumulator())


def test_update_is_cheap_and_rejects_topology_changes(crt):
    hs = crt.HostScene(scene_path("tlas_scene.xml"), 1, ASSETS)
    ctx = crt.Context(64, 64)
    hs.upload(ctx)
    ctx.render(1, 1, 1); ctx.sync()
    T0 = hs.blas_transform(0)[0].reshape(4, 4)
    ts = []
    for f in range(20):
        hs.set_transform(0, rigid(0.1 * f, T0[:3, 3]))
        t0 = time.perf_counter()
        hs.update(ctx, crt.UPDATE_TRANSFORMS)
        ts.append(time.perf_counter() - t0)
    ctx.sync()
    assert sorted(ts)[10] < 1e-3, sorted(ts)                           # < 1 ms for the 3-BLAS scene (host flatten + one async copy of a few KB)
    other = crt.HostScene(scene_path("bunny_scene.xml"), 0, ASSETS)
    with pytest.raises(crt.CrtError):
        other.update(ctx, crt.UPDATE_BOUNDS)                           # a different scene: kind / counts differ
    with pytest.raises(crt.CrtError):
        hs.update(ctx, 8)
    fs = crt.HostScene(scene_path("bunny_scene.xml"), 0, ASSETS)
    c2 = crt.Context(64, 64); fs.upload(c2)
    with pytest.raises(crt.CrtError):
        fs.update(c2, crt.UPDATE_TRANSFORMS)                           # a FileScene bakes its transforms into the triangles


def test_refused_updates_leave_the_scene_intact(crt, orc):
    """a two-level update without the rebuilt TLAS (tlasNodes NULL / wrong count) is CRT_ERR_INVALID for every `what` — not a stale TLAS or a crash —
    and a refused update (permuted triangleIndices) does not touch the host mirror: the next valid update still gives the oracle's image"""
    W, H = 96, 64
    tex = np.full((4, 4), 0x808080, np.uint32)
    ident = np.eye(4, dtype=np.float32)
    # two-level scene, description passed straight through the C ABI
    hs = crt.HostScene(scene_path("tlas_scene.xml"), 1, ASSETS)
    ctx = crt.Context(W, H); hs.upload(ctx); ctx.render(1, 1, 1); ctx.sync()
    bvhs = []
    for i in range(hs.bvh_count()):
        b = hs.bvh(i); T, invT, _, _ = hs.blas_transform(i)
        b.update(objIdx=i + 2, matIdx=0, T=T, invT=invT); bvhs.append(b)
    nodes, used = hs.tlas()
    common = dict(textures=[tex, tex], floor_texture=0, sky_texture=1, materials=[(0.0, 0.0, (0.0, 0.0, 0.0), -1)], light_T=ident, light_invT=ident)
    for what in (crt.UPDATE_BOUNDS, crt.UPDATE_TRANSFORMS, crt.UPDATE_BOUNDS | crt.UPDATE_TRANSFORMS):
        with pytest.raises(crt.CrtError):
            ctx.upload_desc(crt.SCENE_TLAS, bvhs, tlas_nodes=None, update_what=what, **common)
        with pytest.raises(crt.CrtError):
            ctx.upload_desc(crt.SCENE_TLAS, bvhs, tlas_nodes=nodes[:used - 1], update_what=what, **common)
    ctx.upload_desc(crt.SCENE_TLAS, bvhs, tlas_nodes=nodes[:used], update_what=crt.UPDATE_BOUNDS | crt.UPDATE_TRANSFORMS, **common)      # the complete description is accepted
    ctx.clear(); ctx.render(1, 2, 1)
    o, _ = orc.load_scene(scene_path("tlas_scene.xml"), 1, ASSETS); o.renderer_init(W, H); o.render(2, 2)
    assert np.array_equal(ctx.accumulator(), o.accumulator())
    # single-level scene: a permuted leaf order is refused, the mirror is untouched, the next refit is still exact
    from test_oracle_pinning import deform
    fs = crt.HostScene(scene_path("bunny_scene.xml"), 0, ASSETS)
    c2 = crt.Context(W, H); fs.upload(c2); c2.render(1, 1, 1); c2.sync()
    b = fs.bvh(0); bad = dict(b); bad["triIndices"] = b["triIndices"][::-1].copy()
    with pytest.raises(crt.CrtError):
        c2.upload_desc(crt.SCENE_FILE, [bad], obj_mat_idx=[0], update_what=crt.UPDATE_BOUNDS, **common)
    t = b["tris"]; moved = deform(np.stack([t["vertex0"], t["vertex1"], t["vertex2"]], axis=1))
    o2, _ = orc.load_scene(scene_path("bunny_scene.xml"), 0, ASSETS); o2.renderer_init(W, H)
    fs.move_and_refit(0, moved); o2.move_and_refit(0, moved)
    fs.update(c2, crt.UPDATE_BOUNDS)
    c2.clear(); c2.render(1, 2, 1); o2.render(2, 2)
    assert np.array_equal(c2.accumulator(), o2.accumulator())
