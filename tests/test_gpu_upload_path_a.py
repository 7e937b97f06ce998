"""INTEGRATION.md path A: crt_upload_scene fed DIRECTLY (ctypes crt_scene_desc) with arrays the REAL reference built — bvhNodes and
triangleIndices out of infra/bvh.cpp's BVH::Build, compiled where it lies (oracle/_ref), plus its input triangles, committed as
tests/golden/ref_bunny_built.npz — and scene.FindNearest on the GPU checked against the reference's own BVH::Intersect results for the
same rays (tests/golden/ref_bvh_rays.npz).  Nothing of this repo's host front (loaders, builder, scene classes) takes part."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def rigid(t):
    m = np.eye(4, dtype=np.float32); m[:3, 3] = t
    return m


def upload_bunny(crt, ctx):
    z = np.load(os.path.join(GOLDEN, "ref_bunny_built.npz"))
    floor = np.full((512, 512), 0x808080, np.uint32); sky = np.full((4, 8), 0x6080c0, np.uint32)
    bvh = dict(nodes=z["nodes"], tris=z["tris"], triIndices=z["triIndices"])
    # FileScene's light quad and floor plane (file_scene.cpp:15-19): Quad(0, 1) at light_position (0, 3, 1), Plane y = -1
    ctx.upload_desc(crt.SCENE_FILE, [bvh], [floor, sky], 0, 1, [(0.0, 0.0, (0.0, 0.0, 0.0), -1)], rigid((0, 3, 1)), rigid((0, -3, -1)), obj_mat_idx=[0])
    return z


def test_find_nearest_on_reference_built_arrays(crt):
    ctx = crt.Context(64, 64)
    z = upload_bunny(crt, ctx)
    r = np.load(os.path.join(GOLDEN, "ref_bvh_rays.npz"))
    O, D = r["bunny_O"], r["bunny_D"]
    h = ctx.find_nearest(O, D)
    ro, rt = r["bunny_objIdx"], r["bunny_t"]
    both = (h["objIdx"] >= 2) & (ro >= 2)                      # the mesh is the nearest thing in both (FindNearest also tests the light quad and the floor)
    assert both.sum() > 200
    for f in ("t", "u", "v"):
        assert np.array_equal(h[f][both].view(np.uint32), r["bunny_" + f][both].view(np.uint32)), f
    assert np.array_equal(h["triIdx"][both], r["bunny_triIdx"][both])
    other = (ro >= 2) & ~(h["objIdx"] >= 2)
    assert np.all(h["t"][other] < rt[other])
    assert not ((h["objIdx"] >= 2) & (ro < 2)).any()
    miss = (h["objIdx"] == -1) & (ro == -1)
    assert np.array_equal(h["traversed"][miss], r["bunny_traversed"][miss]) and np.array_equal(h["tested"][miss], r["bunny_tested"][miss])


def test_render_on_reference_built_arrays_matches_host_front_upload(crt, tmp_path):
    """the same scene through the repo's own host front (XML + OBJ loader + builder) renders the same image, both render kernels"""
    from test_gpu_golden_and_edges import write_scene
    frames = 70
    a = crt.Context(96, 64); upload_bunny(crt, a); a.render(1, frames, 1)
    # the fixture's triangles are the oracle's scene assembly of bunny.obj at (0,-1,2), rot (0,180,0); flat grey floor / sky as above
    import zlib
    hs = crt.HostScene(write_scene(tmp_path, "bunny"), 0, os.path.join(os.path.dirname(GOLDEN), "..", "assets"))
    b = hs.bvh(0)
    z = np.load(os.path.join(GOLDEN, "ref_bunny_built.npz"))
    assert np.array_equal(b["nodes"], z["nodes"]) and np.array_equal(b["triIndices"], z["triIndices"]) and b["tris"].tobytes() == z["tris"].tobytes()
    acc = a.accumulator()
    assert np.isfinite(acc).all() and acc[..., :3].any()


def test_upload_validation_of_foreign_arrays(crt):
    """negative cases at the C ABI that path A exposes to arrays built elsewhere"""
    z = np.load(os.path.join(GOLDEN, "ref_bunny_built.npz"))
    floor = np.full((512, 512), 0x808080, np.uint32); sky = np.full((4, 8), 0x6080c0, np.uint32)
    ctx = crt.Context(64, 64)
    good = dict(nodes=z["nodes"], tris=z["tris"], triIndices=z["triIndices"])

    def up(bvh, **kw):
        args = dict(textures=[floor, sky], floor_texture=0, sky_texture=1, materials=[(0.0, 0.0, (0, 0, 0), -1)], light_T=rigid((0, 3, 1)), light_invT=rigid((0, -3, -1)), obj_mat_idx=[0])
        args.update(kw)
        ctx.upload_desc(crt.SCENE_FILE, [bvh], **args)

    tris = z["tris"].copy(); tris["objIdx"][17] = 9                 # a triangle whose object has no material entry (checked for EVERY triangle)
    with pytest.raises(crt.CrtError):
        up(dict(good, tris=tris))
    idx = z["triIndices"].copy(); idx[5] = len(idx) + 3               # triangleIndices out of range
    with pytest.raises(crt.CrtError):
        up(dict(good, triIndices=idx))
    nodes = z["nodes"].copy(); nodes["leftFirst"][0] = 2              # children not allocated pairwise
    with pytest.raises(crt.CrtError):
        up(dict(good, nodes=nodes))
    with pytest.raises(crt.CrtError):
        up(good, floor_texture=5)
    up(good)                                                          # and the context is still usable
    assert ctx.find_nearest(np.array([[0, 0, -2]], np.float32), np.array([[0, 0, 1]], np.float32))["objIdx"][0] >= -1
