#!/usr/bin/env python3
"""Generate assets/sky_gradient.png: a deterministic 8-bit equirectangular sky (1024x512) that stands in for the
reference scenes' `industrial_sunset_puresky_4k.hdr`, which is not shipped with the reference checkout
(.MISSING_LARGE_BLOBS).  Pure integer/NumPy arithmetic + zlib, so the file is reproducible byte for byte."""
import os
import struct
import zlib

import numpy as np


def write_png(path, rgb):
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(png)


def main():
    W, H = 1024, 512
    y = np.arange(H, dtype=np.int64)[:, None]          # row 0 = zenith, row H-1 = nadir
    x = np.arange(W, dtype=np.int64)[None, :]
    # vertical gradient: deep blue at the zenith -> pale horizon -> grey-brown ground
    t = np.minimum(y, H // 2) * 255 // (H // 2)        # 0..255 down to the horizon
    r = 40 + t * 170 // 255
    g = 90 + t * 140 // 255
    b = 200 + t * 40 // 255
    ground = y >= H // 2
    gt = (y - H // 2) * 255 // (H // 2)
    r = np.where(ground, 150 - gt * 80 // 255, r)
    g = np.where(ground, 140 - gt * 80 // 255, g)
    b = np.where(ground, 130 - gt * 80 // 255, b)
    r = np.broadcast_to(r, (H, W)).copy()
    g = np.broadcast_to(g, (H, W)).copy()
    b = np.broadcast_to(b, (H, W)).copy()
    # a sun disc and some azimuthal structure so that u (phi) errors are visible too
    sx, sy, rad = 700, 150, 18
    d2 = (x - sx) ** 2 + (y - sy) ** 2
    sun = d2 <= rad * rad
    halo = (d2 <= (4 * rad) ** 2) & ~sun
    r = np.where(sun, 255, np.where(halo, np.minimum(255, r + 40), r))
    g = np.where(sun, 244, np.where(halo, np.minimum(255, g + 30), g))
    b = np.where(sun, 214, b)
    band = ((x // 64) % 2 == 0) & (y > H // 2 - 24) & (y < H // 2)
    r = np.where(band, np.minimum(255, r + 12), r)
    img = np.stack([r, g, b], axis=2).astype(np.uint8)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "sky_gradient.png")
    write_png(out, img)
    print("wrote", out, img.shape)


if __name__ == "__main__":
    main()
