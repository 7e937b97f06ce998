"""Full-size runs (BASELINE.json sizes) checked through size-independent properties + oracle spot checks on a few tiles."""
import numpy as np
import pytest

from conftest import ASSETS, scene_path

pytestmark = pytest.mark.gpu

W, H, SPP = 1280, 720, 64


@pytest.fixture(scope="module")
def full(crt):
    hs = crt.HostScene(scene_path("bunny_scene.xml"), 0, ASSETS)
    ctx = crt.Context(W, H)
    hs.upload(ctx)
    ctx.render(1, SPP, 1)
    return hs, ctx, ctx.accumulator(), ctx.counters()


def test_config2_counters_and_finiteness(full):
    _, _, acc, c = full
    tiles = (W // 16) * (H // 16)
    assert c["primary"] == tiles * 256 * SPP == W * H * SPP
    assert c["primary"] <= c["rays"] <= 6 * c["primary"]
    assert np.isfinite(acc).all() and (acc[..., :3] >= 0).all() and not acc[..., 3].any()
    assert acc[..., :3].max() <= 24.0 * SPP * 1.0001


def test_config2_is_deterministic_and_tile_split_invariant(crt, full):
    """idempotence (same launch again -> same bits) and multi-GPU style tile ownership: 3 interleaved contexts, summed, equal
    the single-context image exactly (each pixel is non-zero in one of them)"""
    hs, _, acc, c = full
    ctx2 = crt.Context(W, H)
    hs.upload(ctx2)
    ctx2.render(1, SPP, 1)
    assert np.array_equal(ctx2.accumulator(), acc)
    ctx2.close()
    tiles = (W // 16) * (H // 16)
    total = np.zeros_like(acc)
    rays = 0
    for r in range(3):
        first, stride, count = crt.tile_partition(r, 3, tiles)
        cx = crt.Context(W, H, tile_first=first, tile_stride=stride, tile_count=count)
        hs.upload(cx)
        cx.render(1, SPP, 1)
        part = cx.accumulator()
        assert not (total.astype(bool) & part.astype(bool)).any()
        total += part
        rays += cx.counters()["rays"]
        cx.close()
    assert np.array_equal(total, acc) and rays == c["rays"]


def test_config2_frame_window_additivity(crt, full):
    """linearity over frames: 64 frames = (spp 1..40) then (spp 41..64) on the same accumulator; and 4 launches of 16 frames"""
    hs, _, acc, _ = full
    cx = crt.Context(W, H)
    hs.upload(cx)
    cx.render(1, 40, 1)
    cx.render(41, 24, 1)
    assert np.array_equal(cx.accumulator(), acc)
    cx.close()
    cx = crt.Context(W, H, max_frames_per_launch=16)
    hs.upload(cx)
    cx.render(1, SPP, 1)
    assert np.array_equal(cx.accumulator(), acc)
    assert cx.timing()["render_launches"] == 4
    cx.close()


def test_config2_spot_tiles_against_oracle(orc, full):
    """the oracle renders 6 of the 3600 tiles at full size (all 64 frames); those pixels must match bit for bit"""
    _, _, acc, _ = full
    o, _ = orc.load_scene(scene_path("bunny_scene.xml"), 0, ASSETS)
    o.renderer_init(W, H)
    tw = W // 16
    tiles = [0, tw * 10 + 3, tw * 31 + 40, tw * 32 + 41, tw * 40 + 20, tw * 44 + 79]
    for t in tiles:
        o.clear()
        o.set_tile_range(t, 1)
        o.render(SPP, 1)
        x0, y0 = (t % tw) * 16, (t // tw) * 16
        want = o.accumulator()[y0:y0 + 16, x0:x0 + 16]
        got = acc[y0:y0 + 16, x0:x0 + 16]
        assert np.abs(got - want).max() / SPP <= 1e-4, t
        assert np.array_equal(got, want), t


def test_config3_tlas_full_size_spot_tiles(crt, orc):
    """BASELINE config 3: TLASFileScene (wok + torii + teapot), 1280x720, 64 spp"""
    hs = crt.HostScene(scene_path("tlas_scene.xml"), 1, ASSETS)
    ctx = crt.Context(W, H)
    hs.upload(ctx)
    ctx.render(1, SPP, 1)
    acc = ctx.accumulator()
    assert np.isfinite(acc).all()
    o, _ = orc.load_scene(scene_path("tlas_scene.xml"), 1, ASSETS)
    o.renderer_init(W, H)
    tw = W // 16
    for t in [tw * 20 + 40, tw * 30 + 25, tw * 33 + 52]:
        o.clear()
        o.set_tile_range(t, 1)
        o.render(SPP, 1)
        x0, y0 = (t % tw) * 16, (t // tw) * 16
        assert np.array_equal(acc[y0:y0 + 16, x0:x0 + 16], o.accumulator()[y0:y0 + 16, x0:x0 + 16]), t


def test_config5_4k_1024spp_eight_way_tile_split(crt, orc):
    """BASELINE config 5 at its stated size on one GPU: TLAS scene, 3840x2160, 1024 spp.  The eight tile owners of an 8-GPU run
    (crt.tile_partition(r, 8, tiles): rank r owns tiles r, r + 8, ...) are rendered one after another and summed — what the RCCL reduce of the
    float4 accumulators does: every pixel is non-zero on exactly one rank — and must equal the whole-image context bit for bit;
    three tiles (sky / floor, instanced meshes) are also checked against the oracle at all 1024 spp."""
    hs = crt.HostScene(scene_path("tlas_scene.xml"), 1, ASSETS)
    Wk, Hk, S = 3840, 2160, 1024
    tiles = (Wk // 16) * (Hk // 16)
    whole = crt.Context(Wk, Hk)
    hs.upload(whole)
    whole.render(1, S, 1)
    acc = whole.accumulator()
    rays = whole.counters()["rays"]
    whole.close()
    assert np.isfinite(acc).all() and not acc[..., 3].any()
    total = np.zeros_like(acc)
    owned_rays = 0
    for r in range(8):
        first, stride, count = crt.tile_partition(r, 8, tiles)
        cx = crt.Context(Wk, Hk, tile_first=first, tile_stride=stride, tile_count=count)
        hs.upload(cx)
        cx.render(1, S, 1)
        part = cx.accumulator()
        owned_rays += cx.counters()["rays"]
        cx.close()
        assert not (total.astype(bool) & part.astype(bool)).any()            # ownership: no pixel is touched by two ranks
        total += part
    assert np.array_equal(total, acc) and owned_rays == rays
    o, _ = orc.load_scene(scene_path("tlas_scene.xml"), 1, ASSETS)
    o.renderer_init(Wk, Hk)
    tw = Wk // 16
    for t in [tw * 3 + 7, tw * 70 + 120, tw * 95 + 131]:
        o.clear()
        o.set_tile_range(t, 1)
        o.render(S, 8)
        x0, y0 = (t % tw) * 16, (t // tw) * 16
        assert np.array_equal(acc[y0:y0 + 16, x0:x0 + 16], o.accumulator()[y0:y0 + 16, x0:x0 + 16]), t


def test_config4_tower_1080p_256spp(crt, orc):
    """BASELINE config 4: watch-tower.obj FileScene + textures, 1920x1080, 256 spp = ONE launch covering 4 windows of 64 frames
    (bit-identical to 4 launches of one window each).
    1080 is not a multiple of 16: SCRHEIGHT/16 truncates (renderer.cpp:151), rows 1072..1079 stay untouched."""
    Wt, Ht, S = 1920, 1080, 256
    hs = crt.HostScene(scene_path("tower_scene.xml"), 0, ASSETS)
    ctx = crt.Context(Wt, Ht)
    hs.upload(ctx)
    ctx.render(1, S, 1)
    acc = ctx.accumulator()
    assert ctx.timing()["render_launches"] == 1
    c = ctx.counters()
    assert c["primary"] == (Wt // 16) * (Ht // 16) * 256 * S
    c1 = crt.Context(Wt, Ht, max_frames_per_launch=64)
    hs.upload(c1)
    c1.render(1, S, 1)
    assert np.array_equal(c1.accumulator(), acc) and c1.timing()["render_launches"] == 4
    c1.close()
    # the same scene file naming the reference's own Wood_Tower_Col.jpg (host loader's JPEG reader = stb_image's texels)
    hj = crt.HostScene(scene_path("tower_scene_jpg.xml"), 0, ASSETS)
    c2 = crt.Context(Wt, Ht)
    hj.upload(c2)
    c2.render(1, 64, 1)
    ctx.clear(); ctx.render(1, 64, 1)
    assert np.array_equal(c2.accumulator(), ctx.accumulator())
    c2.close()
    assert np.isfinite(acc).all() and not acc[1072:].any() and acc[:1072, :, :3].any()
    o, _ = orc.load_scene(scene_path("tower_scene.xml"), 0, ASSETS)
    o.renderer_init(Wt, Ht)
    tw = Wt // 16
    for t in [tw * 5 + 7, tw * 33 + 60, tw * 45 + 58, tw * 66 + 119]:
        o.clear()
        o.set_tile_range(t, 1)
        o.render(S, 1)
        x0, y0 = (t % tw) * 16, (t // tw) * 16
        assert np.array_equal(acc[y0:y0 + 16, x0:x0 + 16], o.accumulator()[y0:y0 + 16, x0:x0 + 16]), t


def test_bench_json_contract():
    """bench.py's one-line JSON: the driver's contract fields + the roofline and cpu_baseline objects (small run of the real script)"""
    import json, os, subprocess, sys
    from conftest import REPO
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--width", "320", "--height", "192"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, CRT_BENCH_CPU_BUDGET_S="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k, t in [("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)]:
        assert isinstance(d[k], t), k
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "Mrays/s" and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    assert rf["traffic"] is None            # the PMC figure is attached only to the workload it was collected on
    assert rf["launches"] in (1, 2) and rf["achieved"] > 0          # (the warm-up launch enters the average when it ran the same kernel as the timed job)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "Mrays/s" and "sample" in cb
    assert d["value"] > cb["value"]        # (a 320x192 job is latency-bound on the GPU; the ratio that matters is the full-size bench line's)


def test_bench_starts_its_own_ranks_for_gpus_2():
    """`bench.py --gpus 2` without an external launcher spawns its two ranks itself (torch.distributed.run on 127.0.0.1) before touching the GPU; both
    ranks share the box's one GPU here and talk gloo (CRT_BENCH_BACKEND), rank 0 prints the one line: tile ownership, ONE collective, rccl_ranks = 2"""
    import json, os, subprocess, sys
    from conftest import REPO
    env = dict(os.environ, CRT_BENCH_BACKEND="gloo", CRT_BENCH_CPU_BUDGET_S="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--width", "320", "--height", "192", "--no-single-render"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["rccl_ranks"] == 2 and d["config"]["collective"] in ("reduce", "all_reduce")
    assert "dealt round-robin over 2 ranks" in d["config"]["workload"] and d["value"] > 0 and "cpu_baseline" not in d
