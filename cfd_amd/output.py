"""Output slices of the solution: the reference's `adjGraphToGrid` / `stringDomain` text form and the image
plots of its (disabled) gnuplot path, without gnuplot.

The reference plots every Property of the z = (nz-1)/2 slice per step as a gnuplot `image` with the default
palette, once with a linear and once with a logarithmic colour axis, 5 pixels per cell, x along the
horizontal axis (src/Main.hs:37-52, :285-294).  write_png reproduces that picture (same orientation, same
palette formulae, autoscaled colour range); write_csv / string_domain give the numbers.
"""
import math
import struct
import zlib

import numpy as np

from .solver import OUTPUT_PROPERTIES, _hs_show_double


def string_domain(rows):
    """stringDomain (src/Main.hs:349-352): every value preceded by a blank, every row by a newline,
    numbers as Haskell's `show` prints them."""
    return "".join("\n" + "".join(" " + _hs_show_double(v) for v in row) for row in rows)


def write_csv(path, rows):
    """One line per x-row, values along y, 17 significant digits (round-trips a double)."""
    a = np.asarray(rows, dtype=np.float64)
    with open(path, "w") as f:
        for row in a:
            f.write(",".join(repr(float(v)) for v in row))
            f.write("\n")


def _palette(g):
    """gnuplot's default `rgbformulae 7,5,15`: r = sqrt(g), g = g^3, b = sin(2 pi g) (clamped to [0, 1])."""
    r = np.sqrt(g)
    gr = g ** 3
    b = np.clip(np.sin(2.0 * math.pi * g), 0.0, 1.0)
    return np.stack([r, gr, b], axis=-1)


def render(rows, log=False, scale=5):
    """(height, width, 3) uint8 image of a slice: x to the right, y upwards, `scale` pixels per cell;
    log = logarithmic colour axis (non-positive and non-finite cells take the lowest colour)."""
    a = np.asarray(rows, dtype=np.float64)
    if a.ndim != 2 or a.size == 0:
        raise ValueError("a slice is a non-empty 2-D array")
    v = a.copy()
    ok = np.isfinite(v)
    if log:
        ok &= v > 0
        v[ok] = np.log10(v[ok])
    if ok.any():
        lo, hi = v[ok].min(), v[ok].max()
    else:
        lo, hi = 0.0, 0.0
    g = np.zeros_like(v)
    if hi > lo:
        g[ok] = (v[ok] - lo) / (hi - lo)
    img = (_palette(np.clip(g, 0.0, 1.0)) * 255.0 + 0.5).astype(np.uint8)      # [x, y, rgb]
    img = np.transpose(img, (1, 0, 2))[::-1]                                   # rows = y from the top down
    return np.repeat(np.repeat(img, scale, axis=0), scale, axis=1)


def write_png(path, rows, log=False, scale=5):
    """8-bit RGB PNG of render(rows, log, scale)."""
    img = np.ascontiguousarray(render(rows, log, scale))
    h, w, _ = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(chunk(b"IEND", b""))


def dump_step(g, directory, step, z_level=None, png=True, csv=True):
    """What the reference's plotting block does for one step (src/Main.hs:285-294): every Property of the
    slice, linear and log-scale pictures under <directory>/<Property>[ LOGSCALE]/<step>.png, plus the numbers
    as <directory>/<Property>/<step>.csv.  `g` is an AdjGraph (or a SlabGraph of an undecomposed run)."""
    import os
    for prp in OUTPUT_PROPERTIES:
        rows = g.property_slice(prp, z_level)
        d = os.path.join(directory, prp)
        os.makedirs(d, exist_ok=True)
        if csv:
            write_csv(os.path.join(d, "%d.csv" % step), rows)
        if png:
            write_png(os.path.join(d, "%d.png" % step), rows, log=False)
            dl = os.path.join(directory, prp + " LOGSCALE")
            os.makedirs(dl, exist_ok=True)
            write_png(os.path.join(dl, "%d.png" % step), rows, log=True)
