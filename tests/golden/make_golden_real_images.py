"""Generates tests/golden/real_image_tiles.npz: 384 x 384 RGB crops of three of the reference's own demo photographs
(datasets/sacre_coeur/mapping/*.jpg -- BASELINE.json configs[0]'s images), so that the GPU tests put REAL image content
(high-frequency texture, edges, sensor noise, JPEG blocking) through the hot path on a box that has no /root/reference.
Run in the build container:  python tests/golden/make_golden_real_images.py
The crops are chosen by gradient energy (the most textured 384 x 384 window on a coarse grid of offsets)."""
import glob
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/datasets/sacre_coeur/mapping"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "real_image_tiles.npz")
TILE = 384


def most_textured_window(rgb):
    g = rgb.astype(np.float32).mean(axis=2)
    gy, gx = np.gradient(g)
    e = gx * gx + gy * gy
    ii = np.pad(e.cumsum(0).cumsum(1), ((1, 0), (1, 0)))
    best, arg = -1.0, (0, 0)
    for y in range(0, g.shape[0] - TILE + 1, 16):
        for x in range(0, g.shape[1] - TILE + 1, 16):
            s = ii[y + TILE, x + TILE] - ii[y, x + TILE] - ii[y + TILE, x] + ii[y, x]
            if s > best:
                best, arg = s, (y, x)
    return arg


def main():
    files = sorted(glob.glob(os.path.join(SRC, "*.jpg")))
    assert len(files) >= 3, "reference demo images not found under " + SRC
    tiles, names, offsets = [], [], []
    for f in (files[0], files[3], files[9]):
        rgb = np.asarray(Image.open(f).convert("RGB"))
        y, x = most_textured_window(rgb)
        tiles.append(rgb[y:y + TILE, x:x + TILE].copy())
        names.append(os.path.basename(f))
        offsets.append((y, x))
    np.savez_compressed(OUT, tiles=np.stack(tiles), names=np.array(names), offsets=np.array(offsets))
    print("wrote", OUT, os.path.getsize(OUT), "bytes", names, offsets)


if __name__ == "__main__":
    main()
