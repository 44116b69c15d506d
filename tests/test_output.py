"""Host-side writers for output slices (no GPU): text form, CSV and PNG."""
import os
import struct
import zlib

import numpy as np

from cfd_amd import output


def test_string_domain_matches_reference_format():
    """stringDomain (src/Main.hs:349-352): newline before every row, blank before every value, Haskell show."""
    assert output.string_domain([[1.0, 0.05], [12345678.0, -0.0]]) == "\n 1.0 5.0e-2\n 1.2345678e7 -0.0"


def test_csv_round_trips(tmp_path):
    rows = np.random.default_rng(1).standard_normal((4, 3)) * 1e-3
    p = tmp_path / "s.csv"
    output.write_csv(str(p), rows)
    back = np.loadtxt(str(p), delimiter=",")
    assert (back.view(np.uint64) == rows.view(np.uint64)).all()


def _read_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, {}
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xFFFFFFFF
        chunks.setdefault(tag, b"")
        chunks[tag] += body
        pos += 12 + n
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[b"IHDR"][:10])
    raw = zlib.decompress(chunks[b"IDAT"])
    img = np.frombuffer(raw, np.uint8).reshape(h, 1 + 3 * w)
    assert (img[:, 0] == 0).all() and depth == 8 and ctype == 2
    return img[:, 1:].reshape(h, w, 3)


def test_png_orientation_palette_and_scale(tmp_path):
    rows = np.zeros((3, 2))          # nx = 3, ny = 2
    rows[2, 1] = 1.0                 # largest value: rightmost column, top row of the picture
    p = tmp_path / "a.png"
    output.write_png(str(p), rows, scale=5)
    img = _read_png(str(p))
    assert img.shape == (10, 15, 3)                    # 5 pixels per cell, x to the right
    assert tuple(img[0, 14]) == (255, 255, 0)          # palette end: sqrt(1), 1^3, sin(2 pi) = 0
    assert tuple(img[9, 0]) == (0, 0, 0)
    assert (img[0:5, 10:15] == img[0, 14]).all()


def test_png_log_scale_ignores_non_positive(tmp_path):
    rows = np.array([[0.0, 1.0], [10.0, 100.0], [-5.0, float("nan")]])
    img = output.render(rows, log=True, scale=1)
    assert img.shape == (2, 3, 3)
    mid = img[1 - 1, 0]              # y = 1 row is at the top: cells (x, 1) = 1.0, 100.0, nan
    assert tuple(img[0, 1]) == (255, 255, 0)           # 100 -> top of the range
    assert tuple(mid) == (0, 0, 0)                     # 1.0 -> bottom of the log range
    assert tuple(img[1, 0]) == (0, 0, 0) and tuple(img[0, 2]) == (0, 0, 0)   # 0 and nan: lowest colour


def test_dump_step_writes_every_property(tmp_path):
    class Fake:
        def property_slice(self, prp, z_level=None):
            return np.arange(6.0).reshape(3, 2) + len(prp)
    output.dump_step(Fake(), str(tmp_path), 7)
    from cfd_amd.solver import OUTPUT_PROPERTIES
    for prp in OUTPUT_PROPERTIES:
        assert os.path.exists(os.path.join(str(tmp_path), prp, "7.csv"))
        assert os.path.exists(os.path.join(str(tmp_path), prp, "7.png"))
        assert os.path.exists(os.path.join(str(tmp_path), prp + " LOGSCALE", "7.png"))
