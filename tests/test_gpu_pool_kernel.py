"""render_pool_kernel (stream pool: more RNG streams than lanes per wavefront) against the oracle and against render_tiles_kernel: partial groups
of streams, passes > 1, two-level scenes, materials, statistics builds, tile ownership.  The back end selects the pool kernel by itself only for
launches many times larger than the machine (crt_render); the small cases here force it (CRT_RENDER_KERNEL=pool_always), one full-size case does not."""
import numpy as np
import pytest

from conftest import ASSETS, scene_path

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("xml,kind,W,H,frames,passes", [("bunny_scene.xml", 0, 64, 48, 130, 1), ("tlas_scene.xml", 1, 64, 48, 200, 2),
                                                         ("tower_scene.xml", 0, 48, 32, 97, 3), ("cube_scene.xml", 0, 32, 32, 300, 1)])
def test_pool_kernel_matches_oracle(crt, orc, monkeypatch, xml, kind, W, H, frames, passes):
    monkeypatch.setenv("CRT_RENDER_KERNEL", "pool_always")
    hs = crt.HostScene(scene_path(xml), kind, ASSETS)
    ctx = crt.Context(W, H, collect_stats=True, max_frames_per_launch=4096)      # statistics build of the pool kernel: all counters
    hs.upload(ctx)
    ctx.render(1, frames, passes)
    acc = ctx.accumulator()
    assert ctx.timing()["render_launches"] == 1
    o, _ = orc.load_scene(scene_path(xml), kind, ASSETS)
    o.renderer_init(W, H)
    o.set_params(5, passes)
    o.render(frames, 4)
    assert np.array_equal(acc, o.accumulator())
    assert ctx.counters() == o.counters()
    px, energy = ctx.resolve_screen(1.0 / (1 + frames * passes))
    assert np.array_equal(px, o.screen()) and np.float32(energy) == np.float32(o.energy())


@pytest.mark.parametrize("xml,kind,W,H,frames", [("bunny_scene.xml", 0, 320, 192, 256), ("tlas_scene.xml", 1, 256, 160, 192)])
def test_pool_and_tiles_kernels_are_bit_identical(crt, monkeypatch, xml, kind, W, H, frames):
    hs = crt.HostScene(scene_path(xml), kind, ASSETS)
    out = {}
    for k in ("tiles", "pool_always"):
        monkeypatch.setenv("CRT_RENDER_KERNEL", k)
        ctx = crt.Context(W, H)
        hs.upload(ctx)
        ctx.render(1, frames, 1)
        out[k] = (ctx.accumulator(), ctx.counters()["rays"])
        ctx.close()
    assert np.array_equal(out["pool_always"][0], out["tiles"][0]) and out["pool_always"][1] == out["tiles"][1]


def test_large_job_selects_the_pool_kernel_and_matches_the_tiles_kernel(crt, monkeypatch):
    """1280x720 x 20 windows = 72 000 (tile, window) pairs: the default choice is the stream pool; same bits as one stream per lane"""
    hs = crt.HostScene(scene_path("bunny_scene.xml"), 0, ASSETS)
    ctx = crt.Context(1280, 720); hs.upload(ctx); ctx.render(1, 20 * 64, 1); a = ctx.accumulator(); ra = ctx.counters()["rays"]; assert ctx.timing()["pool_launches"] == 1; ctx.close()
    monkeypatch.setenv("CRT_RENDER_KERNEL", "tiles")
    ctx = crt.Context(1280, 720); hs.upload(ctx); ctx.render(1, 20 * 64, 1); b = ctx.accumulator(); rb = ctx.counters()["rays"]; assert ctx.timing()["pool_launches"] == 0; ctx.close()
    assert np.array_equal(a, b) and ra == rb


def test_pool_kernel_tile_ownership_and_frame_batches(crt, orc, monkeypatch):
    """two interleaved tile owners, launches of 96 frames (a partial group of streams each), summed == one context == oracle"""
    monkeypatch.setenv("CRT_RENDER_KERNEL", "pool_always")
    W, H, frames = 96, 64, 200
    hs = crt.HostScene(scene_path("bunny_scene.xml"), 0, ASSETS)
    tiles = (W // 16) * (H // 16)
    total = np.zeros((H, W, 4), np.float32)
    for r in range(2):
        first, stride, count = crt.tile_partition(r, 2, tiles)
        ctx = crt.Context(W, H, tile_first=first, tile_stride=stride, tile_count=count, max_frames_per_launch=96)
        hs.upload(ctx)
        ctx.render(1, frames, 1)
        total += ctx.accumulator()
        ctx.close()
    o, _ = orc.load_scene(scene_path("bunny_scene.xml"), 0, ASSETS)
    o.renderer_init(W, H)
    o.render(frames, 4)
    assert np.array_equal(total, o.accumulator())


def test_scene_too_large_for_16_bit_references_falls_back(crt, monkeypatch):
    """render_pool_kernel walks 16-bit node references; a scene that does not fit (here: forced) renders with render_tiles_kernel, same bits"""
    monkeypatch.setenv("CRT_RENDER_KERNEL", "pool_always")
    hs = crt.HostScene(scene_path("bunny_scene.xml"), 0, ASSETS)
    ctx = crt.Context(64, 48); hs.upload(ctx); ctx.render(1, 100, 1); a = ctx.accumulator(); ctx.close()
    monkeypatch.setenv("CRT_DEBUG_NO_REF16", "1")
    ctx = crt.Context(64, 48); hs.upload(ctx); ctx.render(1, 100, 1); b = ctx.accumulator(); ctx.close()
    assert np.array_equal(a, b)


def test_timing_events_stay_bounded_in_a_tick_loop(crt):
    """a host that renders frame after frame and never asks for the timing must not accumulate HIP events (completed launches are folded)"""
    import ctypes as C
    hs = crt.HostScene(scene_path("cube_scene.xml"), 0, ASSETS)
    ctx = crt.Context(32, 32)
    hs.upload(ctx)
    for i in range(600):
        ctx.render(1 + i, 1, 1)
        if i % 100 == 99:
            ctx.sync()
    ctx.sync()
    ctx.L.crt_debug_live_events.restype = C.c_int
    assert ctx.L.crt_debug_live_events(ctx.h) <= 4 * (64 + 600 // 6)          # bounded, far below the 2400 events of 600 launches
    t = ctx.timing()
    assert t["render_launches"] == 600 and t["render_kernel_ms"] > 0


def test_failing_launch_is_reported_and_leaves_the_context_usable(crt, orc, monkeypatch):
    """a render launch the runtime refuses (simulated: CRT_DEBUG_FAIL_LAUNCH) -> CRT_ERR_DEVICE from crt_render, nothing half-recorded:
    timing still answers, the frames rendered before it are in the accumulator, and the same context goes on rendering correctly"""
    hs = crt.HostScene(scene_path("cube_scene.xml"), 0, ASSETS)
    ctx = crt.Context(32, 32)
    hs.upload(ctx)
    ctx.render(1, 1, 1)
    monkeypatch.setenv("CRT_DEBUG_FAIL_LAUNCH", "1")
    with pytest.raises(crt.CrtError) as e:
        ctx.render(2, 1, 1)
    assert e.value.code == -2                                   # CRT_ERR_DEVICE
    t = ctx.timing()
    assert t["render_launches"] == 1 and t["render_kernel_ms"] > 0
    monkeypatch.delenv("CRT_DEBUG_FAIL_LAUNCH")
    ctx.render(2, 1, 1)
    o, _ = orc.load_scene(scene_path("cube_scene.xml"), 0, ASSETS)
    o.renderer_init(32, 32); o.render(2, 1)
    assert np.array_equal(ctx.accumulator(), o.accumulator())


@pytest.mark.parametrize("xml,kind,W,H,frames,passes,force", [("bunny_scene.xml", 0, 160, 96, 320, 1, None), ("bunny_scene.xml", 0, 96, 64, 200, 2, "5"), ("tlas_scene.xml", 1, 96, 64, 256, 1, "23"), ("bunny_scene.xml", 0, 96, 64, 130, 1, "tiles")])
def test_split_jobs_match_the_oracle(crt, orc, monkeypatch, xml, kind, W, H, frames, passes, force):
    """Jobs after the first one know what every tile costs: they dispatch the tiles most expensive first and render the most expensive ones with a concurrent
    render_tiles_kernel launch driven by a block table (abi.cpp plan_job).  Same pixels, same counters, whatever the split."""
    monkeypatch.setenv("CRT_RENDER_KERNEL", "tiles" if force == "tiles" else "pool_always")
    monkeypatch.setenv("CRT_PLAN_NO_TRIAL", "1")                             # (every later job planned; the planned-against-plain trial has its own test below)
    if force == "tiles": monkeypatch.setenv("CRT_SPLIT_FORCE", "7")          # a job below the pool's size: everything through the table, 7 tiles with narrow wavefronts
    elif force is not None: monkeypatch.setenv("CRT_SPLIT_FORCE", force)     # (None: whatever the planner decides for this small image)
    o, _ = orc.load_scene(scene_path(xml), kind, ASSETS)
    o.renderer_init(W, H); o.set_params(5, passes); o.render(frames, 4)
    want = o.accumulator()
    hs = crt.HostScene(scene_path(xml), kind, ASSETS)
    ctx = crt.Context(W, H); hs.upload(ctx)
    splits = []
    for i in range(3):
        ctx.clear(); ctx.reset_counters(); ctx.render(1, frames, passes); ctx.sync()
        assert np.array_equal(ctx.accumulator(), want), i
        assert ctx.counters()["rays"] == o.counters()["rays"]
        splits.append(ctx.timing()["split_launches"])
    assert splits[0] == 0 and (force is None or splits[-1] == 1), splits          # the first job measures, later ones split


def test_planned_jobs_are_checked_against_the_plain_launch(crt, orc, monkeypatch):
    """the plan is a model, so a repeating job shape is tried both ways — planned, then plain — and the faster launch is kept (abi.cpp
    planner_prepare).  Whatever the sequence decides, every launch renders the oracle's pixels; launch 2 is the planned one, launch 3 the plain one."""
    monkeypatch.setenv("CRT_RENDER_KERNEL", "pool_always"); monkeypatch.setenv("CRT_SPLIT_FORCE", "5")
    W, H, frames = 96, 64, 200
    o, _ = orc.load_scene(scene_path("bunny_scene.xml"), 0, ASSETS)
    o.renderer_init(W, H); o.render(frames, 4)
    want = o.accumulator()
    hs = crt.HostScene(scene_path("bunny_scene.xml"), 0, ASSETS)
    ctx = crt.Context(W, H); hs.upload(ctx)
    splits = []
    for i in range(6):
        ctx.clear(); ctx.render(1, frames, 1); ctx.sync()
        assert np.array_equal(ctx.accumulator(), want), i
        splits.append(ctx.timing()["split_launches"])
    assert splits[:3] == [0, 1, 0] and splits[3] == splits[4] == splits[5], splits      # measure, planned, plain, then the winner for good
