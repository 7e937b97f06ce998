#!/usr/bin/env python3
"""Generate assets/textures/Wood_Tower_Col.png from the reference's assets/textures/Wood_Tower_Col.jpg, decoded with the
reference's OWN decoder (lib/stb_image.h compiled in place as part of oracle/_ref).  The host loader of this repo reads
PNG / TGA / PNM but not JPEG; storing stb_image's exact decode as a lossless PNG gives config 4 (watch-tower scene) the same
texels the reference would see.  Authoring container only (needs /root/reference); the PNG is committed as a data fixture."""
import os, struct, sys, zlib
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import orc

def write_png(path, rgb):
    h, w, c = rgb.shape
    assert c == 3
    flat = rgb.reshape(h, w * 3).astype(np.int16)
    sub = flat.copy(); sub[:, 3:] = flat[:, 3:] - flat[:, :-3]                      # PNG filter type 1 (Sub): smaller for photographs
    raw = np.concatenate([np.ones((h, 1), np.uint8), (sub & 255).astype(np.uint8)], axis=1).tobytes()
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))

src = "/root/reference/assets/textures/Wood_Tower_Col.jpg"
img = orc.Ref().image_load(src)
out = os.path.join(REPO, "assets", "textures", "Wood_Tower_Col.png")
write_png(out, img)
back = orc.read_png(out)
assert np.array_equal(back, img)
print("wrote", out, img.shape, os.path.getsize(out), "bytes")
