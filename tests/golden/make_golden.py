#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ (run in the authoring container, where /root/reference is mounted
and oracle/_ref — the reference's own bvh.cpp / tinyobj / stb_image compiled in place — can be built).

Two kinds of vectors, named accordingly:
  ref_*   outputs of the REAL reference code (oracle/_ref).  These pin the oracle (and, on the GPU box, the HIP path).
  orc_*   outputs of the CPU oracle for parts of the path no executable reference exists for (integrator, camera,
          TLAS, scene assembly): regression vectors, "parity unpinned" beyond the restatement itself.
Fixtures are DATA (inputs + expected outputs); no reference source text is stored."""
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import orc  # noqa: E402

A = os.path.join(REPO, "assets")
RA = "/root/reference/assets"
MESHES = ["cube", "bunny", "wok", "teapot", "japanese_torii_gate", "watch-tower", "log_fence"]
IMAGES = ["textures/Stylized_Pavement_basecolor.png", "textures/Stylized_Wood_basecolor.tga", "textures/Defuse_wok.png"]


def crc(a):
    return int(zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff)


def simple_scene(mesh, kind=0, pos=(0, -1, 2), rot=(0, 180, 0), scale=(1, 1, 1)):
    o = orc.Oracle(kind)
    o.set_light_position((0, 3, 1))
    o.set_floor_texture(np.full((512, 512), 0x808080, np.uint32))
    o.set_skydome(np.full((4, 8), 0x6080c0, np.uint32))
    o.add_material()
    o.add_object(orc.read_obj(os.path.join(A, mesh + ".obj")), pos, rot, scale, 0)
    o.build()
    return o


def deform(p):
    """vertex animation stand-in: p + 0.1 * (p.yzx * p.zxy), float32 products and sums only (bit-reproducible everywhere)"""
    p = np.asarray(p, np.float32)
    return (p + np.float32(0.1) * (p[..., [1, 2, 0]] * p[..., [2, 0, 1]])).astype(np.float32)


def main():
    orc.build()
    ref = orc.Ref()
    out = {}
    # --- ref_obj: tinyobj-resolved corners of every mesh -------------------------------------------------------
    out["ref_obj"] = {}
    for m in MESHES:
        pos, nrm, uv = ref.obj_load(os.path.join(RA, m + ".obj"))
        out["ref_obj"][m] = dict(corners=int(pos.shape[0]), pos=crc(pos), nrm=crc(nrm), uv=crc(uv))
    # --- ref_img: stb_image decode, packed as Texture::LoadFromFile does ------------------------------------------
    out["ref_img"] = {}
    for f in IMAGES:
        img = ref.image_load(os.path.join(RA, f))
        out["ref_img"][f] = dict(shape=list(img.shape), packed=crc(orc.pack_rgb(img)))
    # --- ref_bvh: BVH::Build of the reference on each mesh's world-space triangles --------------------------------
    out["ref_bvh"] = {}
    rays = {}
    for m in MESHES:
        o = simple_scene(m)
        tris = o.bvh(0)["tris"]
        h, rb = ref.bvh_build(tris)
        out["ref_bvh"][m] = dict(tris=int(len(tris)), nodesUsed=int(rb["nodesUsed"]), maxDepth=int(rb["maxDepth"]),
                                 nodes=crc(rb["nodes"]), triIndices=crc(rb["triIndices"]), tris_crc=crc(tris))
        if m in ("bunny", "teapot", "cube"):
            rng = np.random.default_rng(1234)
            n = 3000
            O = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
            O[:, 1] = np.abs(O[:, 1]) + 0.2
            T = rng.uniform(-0.8, 0.8, (n, 3)).astype(np.float32) + np.array([0, -0.3, 2], np.float32)
            D = T - O
            D = (D / np.linalg.norm(D, axis=1, keepdims=True)).astype(np.float32)
            hits = ref.bvh_intersect(h, O, D)
            rays[m] = dict(O=O, D=D, t=hits["t"], u=hits["u"], v=hits["v"], objIdx=hits["objIdx"], triIdx=hits["triIdx"],
                           traversed=hits["traversed"], tested=hits["tested"])
        if m == "bunny":
            # --- ref_bunny_built: the arrays the REAL bvh.cpp built (nodes, triangleIndices) + its input triangles, in the reference's layouts:
            # fed to crt_upload_scene directly by tests/test_gpu_upload_path_a.py (INTEGRATION.md path A, no host front of this repo involved)
            np.savez_compressed(os.path.join(HERE, "ref_bunny_built.npz"), nodes=rb["nodes"], triIndices=rb["triIndices"], tris=tris)
        if m in ("bunny", "cube", "teapot"):
            # --- ref_refit: the reference's BVH::Refit (bvh.cpp:26-43) after a deterministic deformation of the vertices ---------
            moved = deform(np.stack([tris["vertex0"], tris["vertex1"], tris["vertex2"]], axis=1))
            out.setdefault("ref_refit", {})[m] = dict(nodes=crc(ref.bvh_move_and_refit(h, moved)), moved=crc(moved))
        ref.bvh_free(h)
    np.savez_compressed(os.path.join(HERE, "ref_bvh_rays.npz"), **{"%s_%s" % (m, k): v for m, d in rays.items() for k, v in d.items()})
    # --- ref_camera: the reference's Camera (template/camera.h, compiled for 1024 x 640): default frustum and SetCameraState -----
    rng = np.random.default_rng(77)
    xy = np.concatenate([rng.uniform(0, [1024, 640], (500, 2)), [[0, 0], [1023.999, 639.999], [512, 320], [0.5, 0.25]]]).astype(np.float32)
    cams = {"default": None, "look": ((0.5, 1.25, -3.0), (0.0, 0.0, 2.0)), "side": ((4.0, 0.5, 2.0), (0.0, -0.5, 2.0))}
    cam_out = {}
    for name, pt in cams.items():
        corners, O, D = ref.camera_rays(xy, pt)
        cam_out[name] = dict(pos_target=pt, corners=crc(corners), O=crc(O), D=crc(D))
    out["ref_camera"] = dict(xy=crc(xy), cams=cam_out)
    np.save(os.path.join(HERE, "ref_camera_xy.npy"), xy)
    # --- ref_texture: Texture::LoadFromFile packing, Texture::Sample and Material::GetAlbedo of the reference -----------------------
    tex = ref.texture_load(os.path.join(RA, "textures/Stylized_Pavement_basecolor.png"))
    uv = np.concatenate([rng.uniform(-0.25, 1.25, (2000, 2)), [[0, 0], [1, 1], [1, 0], [0, 1], [0.5, 0.5], [0.999999, 1e-7]]]).astype(np.float32)
    rgb, alb = ref.texture_sample(tex, uv)
    out["ref_texture"] = dict(file="textures/Stylized_Pavement_basecolor.png", texels=crc(tex), uv=crc(uv), rgb=crc(rgb), albedo=crc(alb),
                              tga=crc(ref.texture_load(os.path.join(RA, "textures/Stylized_Wood_basecolor.tga"))),
                              jpg=crc(ref.texture_load(os.path.join(RA, "textures/Wood_Tower_Col.jpg"))))
    np.save(os.path.join(HERE, "ref_texture_uv.npy"), uv)
    # --- ref_alt: FileScene's alternative accelerators, built and traversed by the REAL infra/kdtree.cpp and infra/grid.cpp (oracle/_ref) --------------
    # same triangles and the same rays as ref_bvh / ref_bvh_rays: structure CRCs + the reference's hits (t, u, v, triIdx, traversed, tested)
    out["ref_alt"] = {}
    alt = {}
    zr = np.load(os.path.join(HERE, "ref_bvh_rays.npz"))
    for m in ("bunny", "teapot", "cube"):
        tris = simple_scene(m).bvh(0)["tris"]
        out["ref_alt"][m] = {}
        for kind in ("kd", "grid"):
            a = ref.alt_accel(kind, tris)
            d = a.dump()
            hh = a.intersect(zr[m + "_O"], zr[m + "_D"])
            a.close()
            if kind == "kd":
                out["ref_alt"][m][kind] = dict(nodes=crc(d["nodes"]), refs=crc(d["refs"]), nodeCount=int(len(d["nodes"])), refCount=int(len(d["refs"])), maxDepth=d["maxDepth"], nodesUsed=d["nodesUsed"])
            else:
                out["ref_alt"][m][kind] = dict(resolution=[int(x) for x in d["resolution"]], cellSize=crc(d["cellSize"]), boundsMin=crc(d["boundsMin"]), boundsMax=crc(d["boundsMax"]),
                                               cellStart=crc(d["cellStart"]), refs=crc(d["refs"]), refCount=int(len(d["refs"])))
            for f in ("t", "u", "v", "objIdx", "triIdx", "traversed", "tested"):
                alt["%s_%s_%s" % (m, kind, f)] = hh[f]
    np.savez_compressed(os.path.join(HERE, "ref_alt_rays.npz"), **alt)
    # --- ref_math: the reference's inline tmplmath.h functions on the path + infra/helper.h's Vertex table ------------------------------
    # math_probe input row = a[3], b[3], angles[3] (radians), s[3]; output row (120 floats) = normalize(a)[3], reflect(a, b)[3], cross(a, b)[3], dot(a, b),
    # mat4::Translate(a), RotateX(angles.x), RotateY(angles.y), RotateZ(angles.z), Scale(s) [16 each], FastInvertedTransformNoScale(RotateY(angles.y) with
    # translation a)[16], aabb after Grow(a), Grow(b), Grow(s): bmin[3], bmax[3], Area(); aabb{a,b}.Grow(aabb{angles,s}): bmin[3], bmax[3], Area()
    rng = np.random.default_rng(2024)
    mi = rng.uniform(-4, 4, (400, 12)).astype(np.float32)
    mi[:40, 6:9] = np.deg2rad(rng.integers(-36, 36, (40, 3)) * 10).astype(np.float32)        # the scene files' whole-degree rotations (x kDeg2Rad happens in the caller)
    mi[40:50, 0:3] = 0; mi[50:60, 3:6] = mi[50:60, 0:3]; mi[60:64] = 0; mi[64:70, 9:12] = 1          # zero vector (normalize -> NaN), degenerate boxes, all zero, unit scale
    mo = ref.math_probe(mi)
    np.savez_compressed(os.path.join(HERE, "ref_math.npz"), inputs=mi, outputs=mo)
    base = rng.uniform(-1, 1, (60, 8)).astype(np.float32)
    v8 = np.concatenate([base, base[::3], base[5:25]])                                              # duplicates in a shuffled order
    v8 = v8[rng.permutation(len(v8))]
    z = v8[:12].copy(); z[:, 1] = 0.0; z2 = z.copy(); z2[:, 1] = -0.0                                 # +0 / -0 compare equal: first occurrence wins
    nanv = v8[3:6].copy(); nanv[:, 4] = np.nan                                                        # a NaN component equals nothing (model.cpp:50 then yields index 0)
    v8 = np.concatenate([v8, z, z2, z, nanv, v8[:7], nanv]).astype(np.float32)
    vi, vu, vh = ref.vertex_dedup(v8)
    np.savez_compressed(os.path.join(HERE, "ref_vertex.npz"), corners=v8, idx=vi, unique=vu, hash=vh)
    out["ref_math"] = dict(rows=int(len(mi)), outputs=crc(mo), vertex_corners=int(len(v8)), vertex_unique=int(len(vu)), vertex_idx=crc(vi))
    # --- orc_render: oracle accumulators of the BASELINE scenes at small sizes (regression vectors) ------------------
    out["orc_render"] = {}
    for name, xml, kind, W, H, frames in [("bunny", "bunny_scene.xml", 0, 96, 64, 4), ("tlas", "tlas_scene.xml", 1, 96, 64, 3),
                                          ("cube", "cube_scene.xml", 0, 64, 48, 4)]:
        o, _ = orc.load_scene(os.path.join(A, "scenes", xml), kind, A)
        o.renderer_init(W, H)
        o.render(frames, 4)
        acc = o.accumulator()
        c = o.counters()
        seeds = [o.tile_seed(frames, t) for t in range((W // 16) * (H // 16))]
        np.save(os.path.join(HERE, "orc_render_%s.npy" % name), acc)
        out["orc_render"][name] = dict(xml=xml, kind=kind, W=W, H=H, frames=frames, acc=crc(acc), counters=c,
                                       energy=float(np.float32(o.energy())), screen=crc(o.screen()), last_frame_tile_seeds=crc(np.array(seeds, np.uint32)))
    # --- orc_whitted: config 1 (cube, Whitted, 640x360) hash ---------------------------------------------------------
    o, _ = orc.load_scene(os.path.join(A, "scenes", "cube_scene.xml"), 0, A)
    o.renderer_init(640, 360)
    o.whitted(4)
    out["orc_whitted_cube_640x360"] = dict(acc=crc(o.accumulator()), screen=crc(o.screen()), counters=o.counters())
    json.dump(out, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "golden.json"))


if __name__ == "__main__":
    main()
