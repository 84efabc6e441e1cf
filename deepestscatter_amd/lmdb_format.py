"""The reference's on-disk dataset without liblmdb: a writer (and a small reader) for the LMDB file layout that
`Dataset` (src/Util/Dataset/Dataset.{h,cpp}) and DeepestScatter_Train/LmdbDataset.py use -- one environment FILE
(MDB_NOSUBDIR), one named database per protobuf message name (Dataset.h:94-98), MDB_INTEGERKEY 4-byte keys
(Dataset.cpp:85, LmdbDataset.py:44), proto3 bytes as values.

LMDB is a third-party dependency of the reference (Dependencies.md) that is absent from the target image (no liblmdb, no
`lmdb` Python module), so nothing here could be checked against the real library: the layout below restates LMDB 0.9's
published data format (mdb.c: MDB_page, MDB_node, MDB_db, MDB_meta; data version 1, 4096-byte pages, little endian) --
two meta pages, a main database whose records are the named databases' MDB_db headers (node flag F_SUBDATA), B+trees of
branch and leaf pages with sorted 16-bit node pointers, overflow pages for values that do not fit a node.  The tests
check the writer against this file's own independent reader and against the structural rules (page flags, sort order,
page accounting), which is consistency, not validation; `tools/flat_to_lmdb.py` prefers the real `lmdb` module where it
exists and says which path it took.
"""
from __future__ import annotations

import struct

PAGE = 4096
P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 0x01, 0x02, 0x04, 0x08
F_BIGDATA, F_SUBDATA = 0x01, 0x02
MDB_INTEGERKEY = 0x08
MDB_MAGIC, MDB_DATA_VERSION = 0xBEEFC0DE, 1
P_INVALID = 0xFFFFFFFFFFFFFFFF
PAGEHDR, NODEHDR = 16, 8
NODEMAX = (((PAGE - PAGEHDR) // 2) & ~1) - 2          # mdb.c: me_nodemax; larger leaf nodes move their data to overflow pages


def _even(n: int) -> int:
    return (n + 1) & ~1


def _db_record(flags: int, depth: int, branch: int, leaf: int, overflow: int, entries: int, root: int, pad: int = 0) -> bytes:
    """MDB_db: md_pad, md_flags, md_depth, md_branch_pages, md_leaf_pages, md_overflow_pages, md_entries, md_root (48 bytes)."""
    return struct.pack("<IHHQQQQQ", pad, flags, depth, branch, leaf, overflow, entries, root)


class _Pages:
    def __init__(self):
        self.pages: list[bytes | None] = [None, None]          # 0 and 1 are the meta pages

    def alloc(self, count: int = 1) -> int:
        n = len(self.pages)
        self.pages.extend([None] * count)
        return n

    def put(self, pgno: int, data: bytes):
        assert len(data) == PAGE
        self.pages[pgno] = data


def _node_page(pgno: int, flags: int, nodes: list[bytes]) -> bytes:
    """A branch or leaf page: header, the node pointers in key order, the nodes packed from the end of the page."""
    page = bytearray(PAGE)
    upper = PAGE
    ptrs = []
    for node in nodes:
        upper -= _even(len(node))
        page[upper:upper + len(node)] = node
        ptrs.append(upper)
    lower = PAGEHDR + 2 * len(nodes)
    assert lower <= upper, "page overflow"
    struct.pack_into("<QHHHH", page, 0, pgno, 0, flags, lower, upper)
    struct.pack_into(f"<{len(ptrs)}H", page, PAGEHDR, *ptrs)
    return bytes(page)


def _build_tree(pages: _Pages, records: list[tuple[bytes, bytes, int]]):
    """records: (key, value, node flags), sorted by the database's key order.  -> (root, depth, branch, leaf, overflow)."""
    if not records:
        return P_INVALID, 0, 0, 0, 0
    overflow_pages = 0
    # ---- leaf level
    level: list[tuple[bytes, int]] = []                          # (first key, pgno)
    nodes: list[bytes] = []
    first_key = None
    used = PAGEHDR

    def flush_leaf():
        nonlocal nodes, first_key, used
        pg = pages.alloc()
        pages.put(pg, _node_page(pg, P_LEAF, nodes))
        level.append((first_key, pg))
        nodes, first_key, used = [], None, PAGEHDR

    for key, value, flags in records:
        if NODEHDR + len(key) + len(value) > NODEMAX:
            count = (PAGEHDR + len(value) + PAGE - 1) // PAGE
            ov = pages.alloc(count)
            blob = bytearray(count * PAGE)
            struct.pack_into("<QHHI", blob, 0, ov, 0, P_OVERFLOW, count)
            blob[PAGEHDR:PAGEHDR + len(value)] = value
            for i in range(count):
                pages.put(ov + i, bytes(blob[i * PAGE:(i + 1) * PAGE]))
            overflow_pages += count
            node = struct.pack("<HHHH", len(value) & 0xFFFF, len(value) >> 16, flags | F_BIGDATA, len(key)) + key + struct.pack("<Q", ov)
        else:
            node = struct.pack("<HHHH", len(value) & 0xFFFF, len(value) >> 16, flags, len(key)) + key + value
        need = _even(len(node)) + 2
        if nodes and used + need > PAGE:
            flush_leaf()
        if first_key is None:
            first_key = key
        nodes.append(node)
        used += need
    flush_leaf()
    leaf_pages, branch_pages, depth = len(level), 0, 1
    # ---- branch levels, bottom up.  The first node of a branch page carries no key (it stands for "everything smaller");
    # every branch page gets at least two nodes (MDB_MINKEYS).
    while len(level) > 1:
        chunks: list[list[tuple[bytes, int]]] = [[]]
        used = PAGEHDR
        for key, child in level:
            size = _even(NODEHDR + (len(key) if chunks[-1] else 0)) + 2
            if chunks[-1] and used + size > PAGE:
                chunks.append([])
                used, size = PAGEHDR, _even(NODEHDR) + 2
            chunks[-1].append((key, child))
            used += size
        if len(chunks) > 1 and len(chunks[-1]) == 1:
            chunks[-1].insert(0, chunks[-2].pop())
        upper_level = []
        for chunk in chunks:
            nodes = []
            for i, (key, child) in enumerate(chunk):
                k = b"" if i == 0 else key
                nodes.append(struct.pack("<HHHH", child & 0xFFFF, (child >> 16) & 0xFFFF, (child >> 32) & 0xFFFF, len(k)) + k)
            pg = pages.alloc()
            pages.put(pg, _node_page(pg, P_BRANCH, nodes))
            upper_level.append((chunk[0][0], pg))
        branch_pages += len(upper_level)
        level = upper_level
        depth += 1
    return level[0][1], depth, branch_pages, leaf_pages, overflow_pages


def write_lmdb(path, tables: dict[str, list[tuple[int, bytes]]], map_size: int = 1 << 30) -> None:
    """One environment file with a named MDB_INTEGERKEY database per table; keys are int32 record ids >= 0."""
    pages = _Pages()
    main_records = []
    for name in sorted(tables, key=lambda s: s.encode()):        # the main database compares names as byte strings
        recs = sorted(tables[name])
        assert all(0 <= k < 1 << 31 for k, _ in recs) and len({k for k, _ in recs}) == len(recs)
        tree = _build_tree(pages, [(struct.pack("<I", k), v, 0) for k, v in recs])
        root, depth, branch, leaf, overflow = tree
        main_records.append((name.encode(), _db_record(MDB_INTEGERKEY, depth, branch, leaf, overflow, len(recs), root), F_SUBDATA))
    root, depth, branch, leaf, overflow = _build_tree(pages, main_records)
    last_pg = len(pages.pages) - 1
    free_db = _db_record(MDB_INTEGERKEY, 0, 0, 0, 0, 0, P_INVALID, pad=PAGE)      # mm_dbs[FREE_DBI].md_pad doubles as the page size
    main_db = _db_record(0, depth, branch, leaf, overflow, len(main_records), root)
    for n in (0, 1):
        meta = bytearray(PAGE)
        struct.pack_into("<QHHHH", meta, 0, n, 0, P_META, 0, 0)
        struct.pack_into("<IIQQ", meta, PAGEHDR, MDB_MAGIC, MDB_DATA_VERSION, 0, max(map_size, (last_pg + 1) * PAGE))
        meta[PAGEHDR + 24:PAGEHDR + 24 + 48] = free_db
        meta[PAGEHDR + 72:PAGEHDR + 72 + 48] = main_db
        struct.pack_into("<QQ", meta, PAGEHDR + 120, last_pg, n)                   # mm_last_pg, mm_txnid (the newer meta wins)
        pages.put(n, bytes(meta))
    with open(path, "wb") as f:
        for p in pages.pages:
            assert p is not None
            f.write(p)


# ---- a reader for the same layout (tests; and to look into a file where liblmdb is not at hand) -----------------------
def _walk(data: bytes, pgno: int, out: list, depth_left: int):
    off = pgno * PAGE
    _, _, flags, lower, _ = struct.unpack_from("<QHHHH", data, off)
    n = (lower - PAGEHDR) // 2
    ptrs = struct.unpack_from(f"<{n}H", data, off + PAGEHDR)
    for p in ptrs:
        lo, hi, nflags, ksize = struct.unpack_from("<HHHH", data, off + p)
        key = data[off + p + NODEHDR:off + p + NODEHDR + ksize]
        if flags & P_BRANCH:
            _walk(data, lo | hi << 16 | nflags << 32, out, depth_left - 1)
        else:
            assert flags & P_LEAF and depth_left == 1
            size = lo | hi << 16
            body = off + p + NODEHDR + ksize
            if nflags & F_BIGDATA:
                ov = struct.unpack_from("<Q", data, body)[0]
                oflags = struct.unpack_from("<H", data, ov * PAGE + 10)[0]
                assert oflags & P_OVERFLOW
                value = data[ov * PAGE + PAGEHDR:ov * PAGE + PAGEHDR + size]
            else:
                value = data[body:body + size]
            out.append((key, value, nflags))


def read_lmdb(path) -> dict[str, list[tuple[int, bytes]]]:
    data = open(path, "rb").read()
    metas = []
    for n in (0, 1):
        magic, version, _, _ = struct.unpack_from("<IIQQ", data, n * PAGE + PAGEHDR)
        assert magic == MDB_MAGIC and version == MDB_DATA_VERSION
        last_pg, txnid = struct.unpack_from("<QQ", data, n * PAGE + PAGEHDR + 120)
        metas.append((txnid, n, last_pg))
    _, n, last_pg = max(metas)
    assert len(data) == (last_pg + 1) * PAGE
    _, _, depth, _, _, _, entries, root = struct.unpack_from("<IHHQQQQQ", data, n * PAGE + PAGEHDR + 72)
    named: list = []
    if root != P_INVALID:
        _walk(data, root, named, depth)
    assert len(named) == entries
    out = {}
    for name, rec, nflags in named:
        assert nflags & F_SUBDATA
        _, flags, depth, _, _, _, count, root = struct.unpack_from("<IHHQQQQQ", rec, 0)
        assert flags & MDB_INTEGERKEY
        rows: list = []
        if root != P_INVALID:
            _walk(data, root, rows, depth)
        assert len(rows) == count
        out[name.decode()] = [(struct.unpack("<I", k)[0], v) for k, v, _ in rows]
    return out
