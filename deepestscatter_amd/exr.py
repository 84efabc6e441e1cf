"""Minimal OpenEXR 2 reader/writer for the one layout the path uses (Camera::saveToDisk, Camera.cpp:149-175):
single part, scan lines, FLOAT channels B, G, R, uncompressed, any line order.  OpenEXR is not installed on
the target image; the C++ host writes the same bytes (host/Exr.h)."""
from __future__ import annotations

import struct

import numpy as np

MAGIC = b"\x76\x2f\x31\x01"


def _attr(name: str, typ: str, value: bytes) -> bytes:
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(value)) + value


def write_exr(path, rgb: np.ndarray, decreasing_y: bool = True) -> None:
    """rgb: float32 [H, W, 3]; pixel (x, y) = rgb[y, x] (row 0 = y 0, as the reference's buffer)."""
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w, _ = rgb.shape
    chl = b""
    for name in (b"B", b"G", b"R"):
        chl += name + b"\0" + struct.pack("<iB3xii", 2, 0, 1, 1)
    chl += b"\0"
    head = MAGIC + struct.pack("<i", 2)
    head += _attr("channels", "chlist", chl)
    head += _attr("compression", "compression", b"\0")
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    head += _attr("dataWindow", "box2i", box) + _attr("displayWindow", "box2i", box)
    head += _attr("lineOrder", "lineOrder", bytes([1 if decreasing_y else 0]))
    head += _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    head += _attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0))
    head += _attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    head += b"\0"
    row_bytes = 3 * w * 4
    chunk = 8 + row_bytes
    start = len(head) + 8 * h
    order = range(h - 1, -1, -1) if decreasing_y else range(h)
    position = {y: k for k, y in enumerate(order)}
    table = b"".join(struct.pack("<Q", start + position[y] * chunk) for y in range(h))
    with open(path, "wb") as f:
        f.write(head + table)
        for y in order:
            f.write(struct.pack("<ii", y, row_bytes))
            f.write(np.ascontiguousarray(rgb[y, :, ::-1].T).tobytes())   # B, G, R planes


def read_exr(path) -> np.ndarray:
    """-> float32 [H, W, 3] (R, G, B).  Only what write_exr / host/Exr.h produce."""
    b = open(path, "rb").read()
    if b[:4] != MAGIC or struct.unpack_from("<i", b, 4)[0] != 2:
        raise ValueError("not a single-part OpenEXR 2 file")
    i = 8
    attrs = {}
    while b[i] != 0:
        j = b.index(b"\0", i)
        name = b[i:j].decode()
        k = b.index(b"\0", j + 1)
        typ = b[j + 1:k].decode()
        n = struct.unpack_from("<i", b, k + 1)[0]
        attrs[name] = (typ, b[k + 5:k + 5 + n])
        i = k + 5 + n
    i += 1
    if attrs["compression"][1] != b"\0":
        raise ValueError("only uncompressed files")
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    names, c = [], attrs["channels"][1]
    p = 0
    while c[p] != 0:
        q = c.index(b"\0", p)
        names.append(c[p:q].decode())
        if struct.unpack_from("<i", c, q + 1)[0] != 2:
            raise ValueError("only FLOAT channels")
        p = q + 1 + 16
    table = struct.unpack_from(f"<{h}Q", b, i)
    out = np.empty((h, w, 3), np.float32)
    for y in range(h):
        off = table[y]
        yy, size = struct.unpack_from("<ii", b, off)
        if yy != y0 + y or size != len(names) * w * 4:
            raise ValueError("corrupt scan line")
        planes = np.frombuffer(b, np.float32, len(names) * w, off + 8).reshape(len(names), w)
        for ci, name in enumerate(names):
            if name in "RGB":
                out[y, :, "RGB".index(name)] = planes[ci]
    return out
