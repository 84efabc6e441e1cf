"""Minimal OpenEXR 2 reader/writer for the one layout the path uses (Camera::saveToDisk, Camera.cpp:149-175):
single part, scan lines, FLOAT channels B, G, R, uncompressed, any line order (the reader also takes HALF channels).  OpenEXR is not installed on
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


def read_exr(path, with_info: bool = False):
    """-> float32 [H, W, 3] (R, G, B); with_info: also {"channels": [(name, "HALF" | "FLOAT")], "line_order": 0 | 1}.
    Single-part scan-line files without compression with HALF or FLOAT channels: what write_exr / host/Exr.h produce,
    and enough of the format to read such a file written by the OpenEXR library itself (tests/golden/)."""
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
    channels, c = [], attrs["channels"][1]
    p = 0
    while c[p] != 0:
        q = c.index(b"\0", p)
        pixel_type, _, xs, ys = struct.unpack_from("<iB3xii", c, q + 1)
        if pixel_type not in (1, 2) or (xs, ys) != (1, 1):
            raise ValueError("only HALF / FLOAT channels without subsampling")
        channels.append((c[p:q].decode(), pixel_type))
        p = q + 1 + 16
    row_bytes = sum(w * (2 if t == 1 else 4) for _, t in channels)
    table = struct.unpack_from(f"<{h}Q", b, i)
    out = np.zeros((h, w, 3), np.float32)
    for y in range(h):
        off = table[y]
        yy, size = struct.unpack_from("<ii", b, off)
        if yy != y0 + y or size != row_bytes:
            raise ValueError("corrupt scan line")
        off += 8
        for name, t in channels:                       # one plane per channel, in the channel list's (alphabetical) order
            plane = np.frombuffer(b, np.float16 if t == 1 else np.float32, w, off)
            off += plane.nbytes
            if name in ("R", "G", "B"):
                out[y, :, "RGB".index(name)] = plane
    if with_info:
        return out, {"channels": [(n, "HALF" if t == 1 else "FLOAT") for n, t in channels], "line_order": attrs["lineOrder"][1][0],
                     "attributes": sorted(attrs)}
    return out
