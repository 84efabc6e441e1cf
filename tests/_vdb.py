"""A minimal, independent OpenVDB FILE WRITER for tests (pure Python + numpy + zlib).

Test infrastructure: it produces the inputs that tests/test_vdb.py feeds to the product's reader
(deepestscatter_amd/host/VdbReader.h via ct_load_vdb).  It was written from the published description of the
format (OpenVDB io/Archive.cc, tree/RootNode.h / InternalNode.h / LeafNode.h writeTopology + writeBuffers,
io/Compression.h writeCompressedValues, zipToStream / bloscToStream, the Blosc 1.x frame and the LZ4 block format),
not from the reader; it shares no code with it.  No OpenVDB or Blosc library exists on the image, so the Blosc
container and the LZ4 encoder below are this file's own (greedy hash matcher); liblz4's runtime library IS there, and
the tests use it to cross-check both directions (the real decoder reads this encoder's streams; the product's reader
reads frames whose LZ4 streams the real encoder wrote).  No file here has been written by OpenVDB or Houdini.

    write_vdb(path, values, active, origin=(ox, oy, oz), ...)
values / active: dense numpy arrays indexed [x, y, z] (float32 / bool) placed at index-space `origin`; every 8^3 leaf
that has an active voxel or a non-background value is stored, everything else is background.  `tiles` adds constant
tiles of the upper tree levels.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

COMPRESS_NONE, COMPRESS_ZIP, COMPRESS_ACTIVE_MASK, COMPRESS_BLOSC = 0, 1, 2, 4


# ---- LZ4 block format: a small greedy encoder ---------------------------------------------------------------
def lz4_encode(data: bytes) -> bytes:
    n = len(data)
    out = bytearray()
    table: dict[bytes, int] = {}
    anchor = i = 0
    limit = n - 12                      # a match may not start within the last 12 bytes ...

    def emit(lit: bytes, match_len: int, offset: int):
        ll, ml = len(lit), (match_len - 4 if match_len else 0)
        out.append((min(ll, 15) << 4) | (min(ml, 15) if match_len else 0))
        if ll >= 15:
            r = ll - 15
            while r >= 255:
                out.append(255)
                r -= 255
            out.append(r)
        out.extend(lit)
        if match_len:
            out.extend(struct.pack("<H", offset))
            if ml >= 15:
                r = ml - 15
                while r >= 255:
                    out.append(255)
                    r -= 255
                out.append(r)

    while i <= limit:
        key = data[i:i + 4]
        cand = table.get(key)
        table[key] = i
        if cand is not None and i - cand <= 65535:
            m = 4
            while i + m < n - 5 and data[cand + m] == data[i + m]:   # ... and the last 5 bytes are literals
                m += 1
            emit(data[anchor:i], m, i - cand)
            i += m
            anchor = i
        else:
            i += 1
    emit(data[anchor:], 0, 0)
    return bytes(out)


# ---- the system's own liblz4 (runtime library only; present on the image), when available: the REAL encoder/decoder ----
def system_lz4():
    """-> (compress(bytes) -> bytes, decompress(bytes, n) -> bytes) bound to liblz4.so.1, or None."""
    import ctypes as C
    try:
        L = C.CDLL("liblz4.so.1")
    except OSError:
        return None
    L.LZ4_compressBound.restype = C.c_int
    L.LZ4_compressBound.argtypes = [C.c_int]
    L.LZ4_compress_default.restype = C.c_int
    L.LZ4_compress_default.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    L.LZ4_decompress_safe.restype = C.c_int
    L.LZ4_decompress_safe.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]

    def compress(data: bytes) -> bytes:
        cap = L.LZ4_compressBound(len(data))
        dst = C.create_string_buffer(cap)
        n = L.LZ4_compress_default(data, dst, len(data), cap)
        assert n > 0
        return dst.raw[:n]

    def decompress(data: bytes, n: int) -> bytes:
        dst = C.create_string_buffer(max(n, 1))
        got = L.LZ4_decompress_safe(data, dst, len(data), n)
        assert got == n, (got, n)
        return dst.raw[:n]
    return compress, decompress


LZ4_ENCODER = None   # set to system_lz4()[0] to have blosc_encode use the real library's streams


def system_blosc():
    """The real Blosc 1.x library where one is installed (a conda environment on the image has libblosc.so.1), called
    the way OpenVDB's bloscToStream calls it (io/Compression.cc: clevel 9, shuffle, typesize sizeof(float), LZ4, one
    block, one thread).  -> compress(bytes) -> frame, or None."""
    import ctypes as C
    import ctypes.util
    lib = None
    for cand in (ctypes.util.find_library("blosc"), "libblosc.so.1", "/opt/conda/lib/libblosc.so.1", "/usr/lib/x86_64-linux-gnu/libblosc.so.1"):
        if not cand:
            continue
        try:
            lib = C.CDLL(cand)
            break
        except OSError:
            continue
    if lib is None:
        return None
    lib.blosc_compress_ctx.restype = C.c_int
    lib.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
    lib.blosc_get_version_string.restype = C.c_char_p

    def compress(data: bytes) -> bytes:
        cap = len(data) + 16                                    # BLOSC_MAX_OVERHEAD
        out = C.create_string_buffer(cap)
        n = lib.blosc_compress_ctx(9, 1, 4, len(data), data, out, cap, b"lz4", len(data), 1)
        assert n > 0, n
        return out.raw[:n]

    compress.version = lib.blosc_get_version_string().decode()
    return compress


BLOSC_ENCODER = None   # set to system_blosc() to have the writer's Blosc frames made by the real library


# ---- Blosc 1.x frame with the LZ4 codec and byte shuffle (what OpenVDB's bloscToStream asks Blosc for) -------------
def blosc_encode(data: bytes, typesize: int = 4, blocksize: int = 0, memcpy: bool = False, shuffle: bool = True) -> bytes:
    n = len(data)
    if memcpy:
        return struct.pack("<BBBBIII", 2, 1, 0x2 | (0x1 if shuffle else 0), typesize, n, n, n + 16) + data
    blocksize = blocksize or n
    nblocks = (n + blocksize - 1) // blocksize
    flags = (0x1 if shuffle else 0) | (1 << 5)          # LZ4 format
    blocks = []
    for b in range(nblocks):
        raw = data[b * blocksize:(b + 1) * blocksize]
        bsize = len(raw)
        leftover = (b == nblocks - 1) and (n % blocksize != 0)
        if shuffle and typesize > 1:
            k = bsize // typesize
            a = np.frombuffer(raw[:k * typesize], np.uint8).reshape(k, typesize)
            raw = a.T.tobytes() + raw[k * typesize:]
        nsplits = typesize if (typesize <= 16 and bsize // typesize >= 128 and not leftover) else 1
        ne = bsize // nsplits
        body = bytearray()
        for j in range(nsplits):
            chunk = raw[j * ne:(j + 1) * ne]
            enc = (LZ4_ENCODER or lz4_encode)(chunk)
            if len(enc) >= ne:                          # stored: a stream as long as its output is a plain copy
                enc = chunk
            body += struct.pack("<i", len(enc)) + enc
        blocks.append(bytes(body))
    header_len = 16 + 4 * nblocks
    starts, pos = [], header_len
    for blk in blocks:
        starts.append(pos)
        pos += len(blk)
    return (struct.pack("<BBBBIII", 2, 1, flags, typesize, n, blocksize, pos) + struct.pack(f"<{nblocks}i", *starts) +
            b"".join(blocks))


def _string(s: str) -> bytes:
    b = s.encode()
    return struct.pack("<I", len(b)) + b


def _mask_bytes(bits: np.ndarray) -> bytes:
    """util/NodeMasks.h: bit n of the mask is bit (n & 63) of 64-bit word n >> 6, little-endian words."""
    return np.packbits(bits.astype(np.uint8), bitorder="little").tobytes()


class _Writer:
    def __init__(self, compression: int, half: bool, version: int, blosc_memcpy: bool, force_all_values: bool):
        self.c, self.half, self.version = compression, half, version
        self.blosc_memcpy, self.force_all = blosc_memcpy, force_all_values

    def scalar(self, v: float) -> bytes:
        # background, root tile values and a node's inactive values are written with sizeof(ValueType) = 4 bytes; a
        # half-float grid truncates them to half PRECISION first (io::truncateRealToHalf) -- only the value arrays are
        # stored as 16-bit halves (RootNode::writeTopology, io::writeCompressedValues)
        return struct.pack("<f", float(np.float16(v)) if self.half else v)

    def data(self, values: np.ndarray) -> bytes:
        raw = values.astype(np.float16 if self.half else np.float32).tobytes()
        if self.c & COMPRESS_BLOSC:
            if len(raw) <= 48:                                       # bloscToStream: too small to bother
                return struct.pack("<q", -len(raw)) + raw
            frame = BLOSC_ENCODER(raw) if BLOSC_ENCODER else blosc_encode(raw, typesize=2 if self.half else 4, memcpy=self.blosc_memcpy)
            return struct.pack("<q", len(frame)) + frame
        if self.c & COMPRESS_ZIP:
            z = zlib.compress(raw)
            if len(z) >= len(raw):
                return struct.pack("<q", -len(raw)) + raw
            return struct.pack("<q", len(z)) + z
        return raw

    def compressed_values(self, values: np.ndarray, mask: np.ndarray, background: float) -> bytes:
        """io::writeCompressedValues for one node: `values` and `mask` flat, in the node's offset order."""
        out = bytearray()
        meta = 6                                                     # NO_MASK_AND_ALL_VALS
        sel = None
        inactive = []
        if (self.c & COMPRESS_ACTIVE_MASK) and not self.force_all and self.version >= 222:
            uniq = np.unique(values[~mask])
            if self.half:
                uniq = np.unique(uniq.astype(np.float16).astype(np.float32))
            if len(uniq) == 0 or (len(uniq) == 1 and uniq[0] == background):
                meta = 0                                             # NO_MASK_OR_INACTIVE_VALS
            elif len(uniq) == 1:
                meta, inactive = 2, [uniq[0]]                        # NO_MASK_AND_ONE_INACTIVE_VAL
            elif len(uniq) == 2 and background in uniq:
                other = uniq[0] if uniq[1] == background else uniq[1]
                meta, inactive = 4, [other]                          # MASK_AND_ONE_INACTIVE_VAL: selection on = background
                sel = (~mask) & (values == background)
            elif len(uniq) == 2:
                meta, inactive = 5, [uniq[0], uniq[1]]               # MASK_AND_TWO_INACTIVE_VALS: selection on = second value
                sel = (~mask) & (values == uniq[1])
        if self.version >= 222:
            out += struct.pack("<b", meta)
        for v in inactive:
            out += self.scalar(float(v))
        if sel is not None:
            out += _mask_bytes(sel)
        if meta != 6:
            out += self.data(values[mask])
        else:
            out += self.data(values)
        return bytes(out)


def write_vdb(path, values: np.ndarray, active: np.ndarray, origin=(0, 0, 0), tiles=(), background: float = 0.0,
              compression: int = COMPRESS_NONE, half: bool = False, version: int = 224, grid_offsets: bool = True,
              name: str = "density", grid_type: str = "Tree_float_5_4_3", blosc_memcpy: bool = False,
              force_all_values: bool = False, transform: str = "UniformScaleMap", extra_grids: int = 0):
    """tiles: iterable of (level, (x, y, z), value, active) with level 1 = an 8^3 tile (a leaf's place), level 2 = a
    128^3 tile, level 3 = a 4096^3 root tile; tile origins must be aligned to their size."""
    w = _Writer(compression, half, version, blosc_memcpy, force_all_values)
    values = np.asarray(values, np.float32)
    active = np.asarray(active, bool)
    ox, oy, oz = origin
    # ---- cut the dense block into 8^3 leaves, keyed by their index-space origin
    leaves = {}
    nx, ny, nz = values.shape
    lo = (ox & ~7, oy & ~7, oz & ~7)
    for lx in range(lo[0], ox + nx, 8):
        for ly in range(lo[1], oy + ny, 8):
            for lz in range(lo[2], oz + nz, 8):
                v = np.full((8, 8, 8), background, np.float32)
                m = np.zeros((8, 8, 8), bool)
                sx0, sy0, sz0 = max(lx, ox), max(ly, oy), max(lz, oz)
                sx1, sy1, sz1 = min(lx + 8, ox + nx), min(ly + 8, oy + ny), min(lz + 8, oz + nz)
                v[sx0 - lx:sx1 - lx, sy0 - ly:sy1 - ly, sz0 - lz:sz1 - lz] = values[sx0 - ox:sx1 - ox, sy0 - oy:sy1 - oy, sz0 - oz:sz1 - oz]
                m[sx0 - lx:sx1 - lx, sy0 - ly:sy1 - ly, sz0 - lz:sz1 - lz] = active[sx0 - ox:sx1 - ox, sy0 - oy:sy1 - oy, sz0 - oz:sz1 - oz]
                if m.any() or (v != background).any():
                    leaves[(lx, ly, lz)] = (v, m)
    tiles = list(tiles)
    for level, o, _, _ in tiles:
        size = {1: 8, 2: 128, 3: 4096}[level]
        assert all(c % size == 0 for c in o), "tile origin not aligned"
        if level == 1:
            assert tuple(o) not in leaves, "a level-1 tile cannot share its place with a leaf"
    # ---- tree structure: root children (4096^3) -> level-2 nodes (128^3) -> leaves / level-1 tiles
    roots: dict = {}
    for key in leaves:
        r = tuple(c & ~4095 for c in key)
        n2 = tuple(c & ~127 for c in key)
        roots.setdefault(r, {}).setdefault(n2, {})
    for level, o, _, _ in tiles:
        if level == 1:
            roots.setdefault(tuple(c & ~4095 for c in o), {}).setdefault(tuple(c & ~127 for c in o), {})
        elif level == 2:
            roots.setdefault(tuple(c & ~4095 for c in o), {})
    root_tiles = [(o, v, a) for level, o, v, a in tiles if level == 3]

    def node_arrays(log2dim, child_total, node_origin, children, node_tiles):
        n = 1 << (3 * log2dim)
        dim = 1 << log2dim
        child_mask = np.zeros(n, bool)
        value_mask = np.zeros(n, bool)
        vals = np.full(n, background, np.float32)
        order = []

        def offset(o):
            x, y, z = ((o[i] - node_origin[i]) >> child_total for i in range(3))
            assert 0 <= x < dim and 0 <= y < dim and 0 <= z < dim
            return (x << (2 * log2dim)) | (y << log2dim) | z
        for o in children:
            child_mask[offset(o)] = True
            vals[offset(o)] = 0.0
            order.append((offset(o), o))
        for o, v, a in node_tiles:
            assert not child_mask[offset(o)]
            vals[offset(o)] = v
            value_mask[offset(o)] = a
        return child_mask, value_mask, vals, [o for _, o in sorted(order)]

    def internal_values(vals, child_mask, value_mask):
        # InternalNode::writeTopology: all NUM_VALUES values since file version 222 (node-mask compression); before that only
        # the childMask.countOff() values of the slots without a child, in slot order
        if version < 222:
            return w.compressed_values(vals[~child_mask], value_mask[~child_mask], background)
        return w.compressed_values(vals, value_mask, background)

    topo = bytearray()
    buffers = bytearray()
    topo += struct.pack("<I", 1)                                     # Tree::writeTopology: buffer count
    topo += w.scalar(background)                                     # RootNode::writeTopology
    topo += struct.pack("<II", len(root_tiles), len(roots))
    for o, v, a in sorted(root_tiles):
        topo += struct.pack("<3i", *o) + w.scalar(v) + struct.pack("<B", 1 if a else 0)
    for r in sorted(roots):                                          # the root's table is a std::map ordered by Coord
        topo += struct.pack("<3i", *r)
        l2_tiles = [(o, v, a) for level, o, v, a in tiles if level == 2 and tuple(c & ~4095 for c in o) == r]
        cm, vm, vals, l2_order = node_arrays(5, 7, r, roots[r].keys(), l2_tiles)
        topo += _mask_bytes(cm) + _mask_bytes(vm) + internal_values(vals, cm, vm)
        for n2 in l2_order:
            kids = [k for k in leaves if tuple(c & ~127 for c in k) == n2]
            l1_tiles = [(o, v, a) for level, o, v, a in tiles if level == 1 and tuple(c & ~127 for c in o) == n2]
            cm2, vm2, vals2, leaf_order = node_arrays(4, 3, n2, kids, l1_tiles)
            topo += _mask_bytes(cm2) + _mask_bytes(vm2) + internal_values(vals2, cm2, vm2)
            for k in leaf_order:
                v, m = leaves[k]
                topo += _mask_bytes(m.reshape(-1))                   # LeafNode::writeTopology: x slowest, z fastest
                buffers += _mask_bytes(m.reshape(-1))                # LeafNode::writeBuffers: the mask again, then the values
                if version < 222:
                    buffers += struct.pack("<3i", *k) + struct.pack("<b", 1)
                buffers += w.compressed_values(v.reshape(-1), m.reshape(-1), background)

    # ---- grid: compression flags, metadata, transform, topology, buffers
    grid = bytearray()
    if version >= 222:
        grid += struct.pack("<I", compression)
    meta = [("class", "string", b"fog volume"), ("name", "string", name.encode()),
            ("file_mem_bytes", "int64", struct.pack("<q", 12345))]
    grid += struct.pack("<I", len(meta))
    for k, t, v in meta:
        grid += _string(k) + _string(t) + struct.pack("<I", len(v)) + v
    vec = lambda a: struct.pack("<3d", *a)
    if transform in ("UniformScaleMap", "ScaleMap"):
        grid += _string(transform) + vec((0.1,) * 3) + vec((0.1,) * 3) + vec((10.0,) * 3) + vec((100.0,) * 3) + vec((5.0,) * 3)
    elif transform in ("UniformScaleTranslateMap", "ScaleTranslateMap"):
        grid += _string(transform) + vec((1.0, 2.0, 3.0)) + vec((0.1,) * 3) + vec((0.1,) * 3) + vec((10.0,) * 3) + vec((100.0,) * 3) + vec((5.0,) * 3)
    elif transform == "AffineMap":
        grid += _string(transform) + struct.pack("<16d", *np.eye(4).reshape(-1))
    else:
        grid += _string(transform)                                    # an unsupported map: the reader must say so
    grid += topo
    block_rel = len(grid)
    grid += buffers

    out = bytearray()
    out += struct.pack("<q", 0x56444220) + struct.pack("<III", version, 9, 0) + struct.pack("<B", 1 if grid_offsets else 0)
    if version < 222:
        out += struct.pack("<B", 1 if compression & COMPRESS_ZIP else 0)
    out += b"01234567-89ab-cdef-0123-456789abcdef"                    # UUID, 36 ASCII characters
    out += struct.pack("<I", 1) + _string("creator") + _string("string") + struct.pack("<I", 7) + b"_vdb.py"
    out += struct.pack("<I", 1 + extra_grids)
    tname = grid_type + ("_HalfFloat" if half else "")
    desc_len = len(_string(name)) + len(_string(tname)) + len(_string("")) + 24
    grid_pos = len(out) + desc_len
    out += _string(name) + _string(tname) + _string("")
    out += struct.pack("<qqq", grid_pos if grid_offsets else 0, grid_pos + block_rel if grid_offsets else 0,
                       grid_pos + len(grid) if grid_offsets else 0)
    out += grid
    for i in range(extra_grids):                                       # further grids the loader must ignore
        out += _string(f"extra{i}") + _string("Tree_vec3s_5_4_3") + _string("") + struct.pack("<qqq", len(out), len(out), len(out))
    with open(path, "wb") as f:
        f.write(out)
    return len(out)


def reference_texture(values: np.ndarray, active: np.ndarray, origin, tiles=(), background: float = 0.0) -> np.ndarray:
    """Resources::loadVolumeBuffer's arithmetic (Resources.cpp:90-141) on the dense arrays themselves -- no tree, no file:
    uint8 [Z, Y, X] texture of the active bounding box expanded by one voxel."""
    values = np.asarray(values, np.float32)
    active = np.asarray(active, bool)
    boxes = []
    if active.any():
        idx = np.argwhere(active)
        boxes.append((idx.min(0) + origin, idx.max(0) + origin))
    vmax = -np.inf if not active.any() else float(values[active].max())
    size = {1: 8, 2: 128, 3: 4096}
    for level, o, v, a in tiles:
        if a:
            boxes.append((np.array(o), np.array(o) + size[level] - 1))
            vmax = max(vmax, float(np.float32(v)))
    lo = np.min([b[0] for b in boxes], axis=0) - 1
    hi = np.max([b[1] for b in boxes], axis=0) + 1
    dims = hi - lo + 1
    dense = np.full(tuple(dims), background, np.float32)               # [x, y, z]

    def paste(block, at):
        a0 = np.maximum(at, lo)
        a1 = np.minimum(at + np.array(block.shape), hi + 1)
        if (a1 > a0).all():
            dense[a0[0] - lo[0]:a1[0] - lo[0], a0[1] - lo[1]:a1[1] - lo[1], a0[2] - lo[2]:a1[2] - lo[2]] = \
                block[a0[0] - at[0]:a1[0] - at[0], a0[1] - at[1]:a1[1] - at[1], a0[2] - at[2]:a1[2] - at[2]]
    for level, o, v, a in sorted(tiles, key=lambda t: -t[0]):         # larger tiles first, smaller ones on top
        paste(np.full((size[level],) * 3, v, np.float32), np.array(o))
    # leaves win over tiles: every 8^3 leaf the writer stores
    ox, oy, oz = origin
    nx, ny, nz = values.shape
    for lx in range(ox & ~7, ox + nx, 8):
        for ly in range(oy & ~7, oy + ny, 8):
            for lz in range(oz & ~7, oz + nz, 8):
                v = np.full((8, 8, 8), background, np.float32)
                m = np.zeros((8, 8, 8), bool)
                s0 = np.maximum((lx, ly, lz), (ox, oy, oz))
                s1 = np.minimum((lx + 8, ly + 8, lz + 8), (ox + nx, oy + ny, oz + nz))
                sl = tuple(slice(s0[i] - (lx, ly, lz)[i], s1[i] - (lx, ly, lz)[i]) for i in range(3))
                sr = tuple(slice(s0[i] - (ox, oy, oz)[i], s1[i] - (ox, oy, oz)[i]) for i in range(3))
                v[sl] = values[sr]
                m[sl] = active[sr]
                if m.any() or (v != background).any():
                    paste(v, np.array((lx, ly, lz)))
    q = (dense.astype(np.float64) / vmax * 255)                         # float / double * int, truncated to uint8
    return np.ascontiguousarray(q.astype(np.uint8).transpose(2, 1, 0))
