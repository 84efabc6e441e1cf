"""The built-in LMDB writer (deepestscatter_amd/lmdb_format.py; SURVEY section 8 f-2: the reference's on-disk layout,
Dataset.cpp:13-17,85) against its own reader and against the structural rules of the format.  liblmdb is not on the
image: this is consistency, not validation against the real library (the module's header says so)."""
import struct
import subprocess
import sys
from pathlib import Path

import numpy as np

from deepestscatter_amd import collector as col
from deepestscatter_amd import lmdb_format as L

ROOT = Path(__file__).resolve().parents[1]


def _tables(n_results, n_desc):
    rng = np.random.default_rng(5)
    results = [(4096 + i, col.encode_result(float(rng.random()), True)) for i in range(n_results)]
    samples = [(4096 + i, col.encode_scatter_sample(2, rng.normal(size=3), rng.normal(size=3))) for i in range(n_results)]
    desc = [(4096 + i, col.encode_disney_descriptor(rng.integers(0, 256, 2250).astype(np.uint8))) for i in range(n_desc)]
    setup = [(2, b"\x0a\x05a.vdb\x15\x00\xc0\xda\x45")]
    return {"Result": results, "ScatterSample": samples, "DisneyDescriptor": desc, "SceneSetup": setup}


def test_round_trip_through_the_builtin_writer_and_reader(tmp_path):
    tables = _tables(2048, 300)                              # one batch of the reference (Tasks.cpp:137); descriptors overflow a node
    path = tmp_path / "Dataset.lmdb"
    L.write_lmdb(path, tables)
    got = L.read_lmdb(path)
    assert got == {k: sorted(v) for k, v in tables.items()}


def test_structure_of_the_file(tmp_path):
    tables = _tables(5000, 40)                               # enough records for two levels of branch pages? (one at least)
    path = tmp_path / "d.lmdb"
    L.write_lmdb(path, tables)
    data = path.read_bytes()
    assert len(data) % L.PAGE == 0
    n_pages = len(data) // L.PAGE
    # meta pages: magic, version, page size in the free DB's md_pad, INTEGERKEY free DB without a root, the newer txnid on page 1
    for n in (0, 1):
        pgno, _, flags, _, _ = struct.unpack_from("<QHHHH", data, n * L.PAGE)
        assert (pgno, flags) == (n, L.P_META)
        magic, version, address, mapsize = struct.unpack_from("<IIQQ", data, n * L.PAGE + 16)
        assert (magic, version, address) == (0xBEEFC0DE, 1, 0) and mapsize >= len(data)
        pad, fl, depth, *_, root = struct.unpack_from("<IHHQQQQQ", data, n * L.PAGE + 16 + 24)
        assert (pad, fl, depth, root) == (4096, L.MDB_INTEGERKEY, 0, L.P_INVALID)
        last_pg, txnid = struct.unpack_from("<QQ", data, n * L.PAGE + 16 + 120)
        assert last_pg == n_pages - 1 and txnid == n
    # every page is numbered as it lies in the file and is a branch, leaf or overflow page; node pointers stay inside the page
    # and keys on a page are in ascending order; page counts add up to what the database headers say
    kinds = {L.P_BRANCH: 0, L.P_LEAF: 0, L.P_OVERFLOW: 0}
    p = 2
    while p < n_pages:
        pgno, _, flags, lower, upper = struct.unpack_from("<QHHHH", data, p * L.PAGE)
        assert pgno == p
        if flags == L.P_OVERFLOW:
            count = struct.unpack_from("<I", data, p * L.PAGE + 12)[0]
            kinds[L.P_OVERFLOW] += count
            p += count
            continue
        assert flags in (L.P_BRANCH, L.P_LEAF) and 16 <= lower <= upper <= L.PAGE
        kinds[flags] += 1
        ptrs = struct.unpack_from(f"<{(lower - 16) // 2}H", data, p * L.PAGE + 16)
        assert all(upper <= q < L.PAGE and q % 2 == 0 for q in ptrs)
        if flags == L.P_BRANCH:
            assert len(ptrs) >= 2
            assert struct.unpack_from("<H", data, p * L.PAGE + ptrs[0] + 6)[0] == 0        # the first node carries no key
        keys = []
        for q in ptrs[1 if flags == L.P_BRANCH else 0:]:
            ksize = struct.unpack_from("<H", data, p * L.PAGE + q + 6)[0]
            keys.append(data[p * L.PAGE + q + 8:p * L.PAGE + q + 8 + ksize])
        if keys and all(len(k) == 4 for k in keys):
            ints = [struct.unpack("<I", k)[0] for k in keys]
            assert ints == sorted(ints) and len(set(ints)) == len(ints)
        else:
            assert keys == sorted(keys)
        p += 1
    main = struct.unpack_from("<IHHQQQQQ", data, L.PAGE + 16 + 72)
    named = L.read_lmdb(path)
    assert main[6] == len(named) == 4
    assert kinds[L.P_OVERFLOW] == 40 and kinds[L.P_BRANCH] >= 2 and kinds[L.P_LEAF] + kinds[L.P_BRANCH] + 40 + 2 == n_pages


def test_converter_tool_without_the_lmdb_module(tmp_path):
    recs = [(i, col.encode_result(0.5 * i, True)) for i in range(50)]
    col.write_flat_dataset(tmp_path / "Result.flat", "Result", recs)
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "flat_to_lmdb.py"), "--builtin", str(tmp_path / "T.lmdb"), str(tmp_path / "Result.flat")],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "built-in writer" in r.stdout, r.stdout + r.stderr
    assert L.read_lmdb(tmp_path / "T.lmdb") == {"Result": recs}
