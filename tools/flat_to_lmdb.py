#!/usr/bin/env python3
"""Convert flat record files (deepestscatter_amd.collector.write_flat_dataset, `cloudtrace collect`) into the reference's
LMDB layout (src/Util/Dataset/Dataset.cpp:13-17,85: MDB_NOSUBDIR | MDB_NOTLS | MDB_WRITEMAP environment, one named DB per
protobuf message name, MDB_INTEGERKEY 4-byte keys), so DeepestScatter_Train/LmdbDataset.py reads our samples unchanged.

With the `lmdb` Python module (where the trainer runs) the real library writes the file.  Without it (the build image)
deepestscatter_amd/lmdb_format.py writes the same layout from the format's description -- consistent with its own reader,
never opened by liblmdb: say so when you hand such a file on.  --builtin forces that path.

    python tools/flat_to_lmdb.py [--builtin] Train.lmdb ScatterSample.flat Result.flat ...
"""
import struct
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    from deepestscatter_amd.collector import read_flat_dataset
    args = [a for a in sys.argv[1:] if a != "--builtin"]
    if len(args) < 2:
        raise SystemExit(__doc__)
    dst, sources = args[0], args[1:]
    tables = {}
    for src in sources:
        table, records = read_flat_dataset(src)
        tables.setdefault(table, []).extend(records)
    lmdb = None
    if "--builtin" not in sys.argv:
        try:
            import lmdb
        except ImportError:
            lmdb = None
    if lmdb is not None:
        env = lmdb.open(dst, subdir=False, max_dbs=64, map_size=1 << 34, writemap=True, lock=True)
        for table, records in tables.items():
            db = env.open_db(table.encode(), integerkey=True)
            with env.begin(write=True, db=db) as txn:
                for key, val in records:
                    txn.put(struct.pack("<i", key), val)
        env.close()
        how = "liblmdb"
    else:
        from deepestscatter_amd.lmdb_format import write_lmdb
        write_lmdb(dst, tables)
        how = "the built-in writer (no lmdb module here: layout from the format description, not opened by liblmdb)"
    print(f"wrote {', '.join(f'{len(r)} {t}' for t, r in tables.items())} records to {dst} with {how}")


if __name__ == "__main__":
    main()
