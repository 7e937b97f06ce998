#!/usr/bin/env python3
"""Generates tests/golden/jpeg/*.jpg + tests/golden/jpeg_golden.json (authoring container only: needs Pillow to ENCODE the small
synthetic test images and oracle/_ref — the reference's own lib/stb_image.h compiled in place — to decode them).

The fixtures pin the host loader's baseline-JPEG decoder (cpu-ray-tracer_amd/csrc/host/loaders.cpp) to the texels the reference
gets from stbi_load (template/texture.h:18): 4:4:4 / 4:2:2 / 4:2:0 sampling, greyscale, odd sizes, restart intervals, progressive scans, and the
reference's own Wood_Tower_Col.jpg.  Fixtures are DATA (JPEG inputs made here + CRCs of the real stb decode)."""
import io
import json
import os
import sys
import zlib

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import orc  # noqa: E402


def crc(a):
    return int(zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xffffffff)


def picture(w, h, seed):
    """smooth gradients + a few hard edges + noise: exercises every coefficient band and the chroma filters"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([128 + 100 * np.sin(x / 5.0) * np.cos(y / 7.0), 255.0 * x / max(w - 1, 1), 255.0 * y / max(h - 1, 1)], axis=2)
    img[h // 3:h // 2, w // 4:w // 2] = (250, 10, 30)
    img[:, w // 2:w // 2 + 2] = (0, 255, 0)
    img += rng.normal(0, 12, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    orc.build()
    ref = orc.Ref()
    out = {}
    os.makedirs(os.path.join(HERE, "jpeg"), exist_ok=True)
    cases = [("rgb444_67x45_q90", (67, 45), "RGB", dict(quality=90, subsampling=0)),
             ("rgb422_67x45_q75", (67, 45), "RGB", dict(quality=75, subsampling=1)),
             ("rgb420_67x45_q85", (67, 45), "RGB", dict(quality=85, subsampling=2)),
             ("rgb420_16x16_q50", (16, 16), "RGB", dict(quality=50, subsampling=2)),
             ("rgb420_1x1_q90", (1, 1), "RGB", dict(quality=90, subsampling=2)),
             ("rgb420_130x3_q95", (130, 3), "RGB", dict(quality=95, subsampling=2)),
             ("rgb422_3x70_q60", (3, 70), "RGB", dict(quality=60, subsampling=1)),
             ("grey_50x33_q80", (50, 33), "L", dict(quality=80)),
             ("rgb420_restart_96x80_q70", (96, 80), "RGB", dict(quality=70, subsampling=2, restart_marker_blocks=3)),
             ("rgb444_restart_40x40_q30", (40, 40), "RGB", dict(quality=30, subsampling=0, restart_marker_rows=1)),
             ("rgb420_optimized_64x48_q88", (64, 48), "RGB", dict(quality=88, subsampling=2, optimize=True)),
             ("prog_rgb420_67x45_q85", (67, 45), "RGB", dict(quality=85, subsampling=2, progressive=True)),
             ("prog_rgb444_40x56_q92", (40, 56), "RGB", dict(quality=92, subsampling=0, progressive=True)),
             ("prog_rgb422_33x17_q40", (33, 17), "RGB", dict(quality=40, subsampling=1, progressive=True)),
             ("prog_grey_50x33_q80", (50, 33), "L", dict(quality=80, progressive=True)),
             ("prog_rgb420_restart_96x80_q70", (96, 80), "RGB", dict(quality=70, subsampling=2, progressive=True, restart_marker_blocks=2)),
             ("prog_rgb420_8x8_q10", (8, 8), "RGB", dict(quality=10, subsampling=2, progressive=True))]
    for i, (name, (w, h), mode, kw) in enumerate(cases):
        px = picture(w, h, 100 + i)
        im = Image.fromarray(px if mode == "RGB" else px[:, :, 0], mode)
        path = os.path.join(HERE, "jpeg", name + ".jpg")
        im.save(path, "JPEG", **kw)
        dec = ref.image_load(path)                       # the REAL stb_image
        out[name] = dict(shape=list(dec.shape), packed=crc(orc.pack_rgb(dec)))
    # 4-component (CMYK) file: the loader must refuse it with a message (stb would convert it; documented limit)
    Image.fromarray(picture(32, 32, 7), "RGB").convert("CMYK").save(os.path.join(HERE, "jpeg", "cmyk_32x32.jpg"), "JPEG", quality=80)
    # the reference's own JPEG texture (BASELINE config 4)
    dec = ref.image_load("/root/reference/assets/textures/Wood_Tower_Col.jpg")
    out["Wood_Tower_Col"] = dict(shape=list(dec.shape), packed=crc(orc.pack_rgb(dec)))
    json.dump(out, open(os.path.join(HERE, "jpeg_golden.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
