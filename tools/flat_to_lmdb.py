#!/usr/bin/env python3
"""Convert a flat record file written by deepestscatter_amd.collector.write_flat_dataset into the
reference's LMDB layout (src/Util/Dataset/Dataset.cpp:13-17,85: MDB_NOSUBDIR | MDB_NOTLS |
MDB_WRITEMAP environment, one named DB per protobuf message name, MDB_INTEGERKEY 4-byte keys), so
DeepestScatter_Train/LmdbDataset.py reads our radiance samples unchanged.  Needs the `lmdb` Python
module, which is not on the build image -- run it where the trainer runs.

    python tools/flat_to_lmdb.py results.flat Train.lmdb
"""
import struct
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    import lmdb  # noqa: imported late so that --help works without it
    from deepestscatter_amd.collector import read_flat_dataset
    src, dst = sys.argv[1], sys.argv[2]
    table, records = read_flat_dataset(src)
    env = lmdb.open(dst, subdir=False, max_dbs=64, map_size=1 << 34, writemap=True, lock=True)
    db = env.open_db(table.encode(), integerkey=True)
    with env.begin(write=True, db=db) as txn:
        for key, val in records:
            txn.put(struct.pack("<i", key), val)
    env.close()
    print(f"wrote {len(records)} {table} records to {dst}")


if __name__ == "__main__":
    main()
