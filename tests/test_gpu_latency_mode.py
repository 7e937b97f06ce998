"""Latency mode of single-window launches (crt_render of 8 .. 64 frames): the back end measures what every tile costs and hands expensive tiles to several
narrower wavefronts (block tables, abi.cpp next_block_table), stage by stage over the caller's first launches.  Whatever table a launch runs with, the pixels
must be the ones of the reference order: every launch of the sequence is compared with the oracle."""
import numpy as np
import pytest

from conftest import ASSETS, scene_path

pytestmark = pytest.mark.gpu


def _oracle(orc, xml, kind, W, H, frames, passes, cam=None):
    o, _ = orc.load_scene(scene_path(xml), kind, ASSETS)
    o.renderer_init(W, H)
    if cam is not None: o.set_camera_state(*cam)
    o.set_params(5, passes)
    o.render(frames, 4)
    return o.accumulator()


@pytest.mark.parametrize("xml,kind,W,H,frames,passes", [("bunny_scene.xml", 0, 160, 96, 64, 1), ("tlas_scene.xml", 1, 96, 64, 40, 2), ("cube_scene.xml", 0, 64, 64, 9, 1)])
def test_every_stage_of_the_latency_mode_renders_the_same_pixels(crt, orc, monkeypatch, xml, kind, W, H, frames, passes):
    monkeypatch.setenv("CRT_RENDER_KERNEL", "tiles")
    want = _oracle(orc, xml, kind, W, H, frames, passes)
    hs = crt.HostScene(scene_path(xml), kind, ASSETS)
    ctx = crt.Context(W, H); hs.upload(ctx)
    for i in range(12):                                 # the probed (or one-wave) launch, the table stages, two confirmation launches, then the fastest table
        ctx.clear(); ctx.render(1, frames, passes); ctx.sync()
        assert np.array_equal(ctx.accumulator(), want), "launch %d of the sequence differs" % i
        if i % 2: ctx.timing()                          # (a caller that reads the timing recycles the launches' event pairs: the tuner has looked at them before)


def test_fixed_tables_down_to_one_lane_per_wavefront(crt, orc, monkeypatch):
    W, H, frames = 96, 64, 64
    want = _oracle(orc, "bunny_scene.xml", 0, W, H, frames, 1)
    hs = crt.HostScene(scene_path("bunny_scene.xml"), 0, ASSETS)
    monkeypatch.setenv("CRT_LAT_FORCE", "1")
    for policy in ("0:1", "0.5:2,0:32", "0.9:4,0.5:16"):         # every tile as 64 one-lane wavefronts; halves; a mix
        monkeypatch.setenv("CRT_LAT_POLICY", policy)
        ctx = crt.Context(W, H); hs.upload(ctx)
        for i in range(4):
            ctx.clear(); ctx.render(1, frames, 1); ctx.sync()
            assert np.array_equal(ctx.accumulator(), want), (policy, i)
        ctx.close()


def test_camera_change_and_frame_count_change_restart_the_measurement(crt, orc, monkeypatch):
    W, H = 96, 64
    hs = crt.HostScene(scene_path("bunny_scene.xml"), 0, ASSETS)
    ctx = crt.Context(W, H, tile_first=1, tile_stride=2, tile_count=12); hs.upload(ctx)      # a strided subset of the tiles, as one rank of a multi-GPU job owns
    owned = np.zeros((H // 16) * (W // 16), bool); owned[1:24:2] = True
    mask = np.repeat(np.repeat(owned.reshape(H // 16, W // 16), 16, axis=0), 16, axis=1)
    cams = [((0.0, 0.0, -2.0), (0.0, 0.0, -1.0)), ((0.6, 0.3, -1.5), (0.0, -0.4, 1.0))]
    for k, cam in enumerate(cams):
        ctx.set_camera_state(cam[0], cam[1])
        for frames in (48, 64, 16, 64, 64, 33, 64):
            want = _oracle(orc, "bunny_scene.xml", 0, W, H, frames, 1, cam)
            ctx.clear(); ctx.render(1, frames, 1); ctx.sync()
            got = ctx.accumulator()
            assert np.array_equal(got[mask], want[mask]) and not got[~mask].any(), (k, frames)


def test_mixed_sequence_of_launch_shapes_and_camera_moves(crt, orc):
    """The scheduling layer keeps state between calls (tile costs, block tables, plans, the tuner's stages): a caller that mixes single-window renders, jobs of
    different lengths and passes, camera moves and timing reads must get the reference's pixels from every call."""
    W, H = 96, 64
    hs = crt.HostScene(scene_path("bunny_scene.xml"), 0, ASSETS)
    o, _ = orc.load_scene(scene_path("bunny_scene.xml"), 0, ASSETS)
    o.renderer_init(W, H)
    ctx = crt.Context(W, H); hs.upload(ctx)
    rng = np.random.default_rng(7)
    cams = [((0.0, 0.0, -2.0), (0.0, 0.0, -1.0)), ((0.5, 0.2, -1.6), (0.0, -0.3, 1.0)), ((-0.7, 0.1, -1.2), (0.0, -0.5, 1.5))]
    cam = cams[0]
    for step in range(26):
        if step in (9, 17): cam = cams[1 + (step > 9)]; ctx.set_camera_state(*cam)
        frames = int(rng.choice([1, 7, 16, 64, 64, 64, 130, 192, 256, 320]))
        passes = int(rng.choice([1, 1, 1, 2]))
        ctx.clear(); ctx.render(1, frames, passes); ctx.sync()
        if step % 5 == 4: ctx.timing()
        o.set_camera_state(*cam); o.clear(); o.set_spp(1); o.set_params(5, passes); o.render(frames, 4)
        assert np.array_equal(ctx.accumulator(), o.accumulator()), (step, frames, passes)


@pytest.mark.parametrize("xml,kind,lanes", [("bunny_scene.xml", 0, 1), ("bunny_scene.xml", 0, 2), ("tlas_scene.xml", 1, 2)])
def test_narrow_kernel_renders_the_same_pixels(crt, orc, monkeypatch, xml, kind, lanes):
    """render_narrow_kernel (opt-in: CRT_NARROW_LANES) takes the one- / two-lane blocks of a table: plain per-lane path loops, both children's records requested
    ahead, record stack and treetop in LDS — the pixels must be the oracle's whatever share of the tiles it renders"""
    W, H, frames = 96, 64, 64
    want = _oracle(orc, xml, kind, W, H, frames, 1)
    hs = crt.HostScene(scene_path(xml), kind, ASSETS)
    monkeypatch.setenv("CRT_RENDER_KERNEL", "tiles")
    monkeypatch.setenv("CRT_NARROW_LANES", str(lanes))
    monkeypatch.setenv("CRT_LAT_FORCE", "1")
    for policy in ("0:1", "0.5:2,0:32", "0.6:1,0.3:2"):
        monkeypatch.setenv("CRT_LAT_POLICY", policy)
        ctx = crt.Context(W, H); hs.upload(ctx)
        for i in range(4):
            ctx.clear(); ctx.render(1, frames, 1); ctx.sync()
            assert np.array_equal(ctx.accumulator(), want), (policy, i)
        ctx.close()
