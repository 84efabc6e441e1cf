#!/usr/bin/env python3
"""Convert the first FloatGrid of a .vdb file to the `.f32grid` volume the headless host reads
(deepestscatter_amd/host/Scene.h: int32 nx, ny, nz + nx*ny*nz float32, x fastest), with the reference
loader's semantics (Resources::loadVolumeBuffer, Resources.cpp:82-143): active bounding box expanded by
one voxel on every side, dense fill, values as they are (the uint8 quantisation `trunc(v / max * 255)` is
done by ct_quantize_volume when the volume is loaded).

Needs the `pyopenvdb` (or `openvdb`) Python module, which is NOT on the MI355X image: run it wherever the
.vdb files are produced (Houdini ships one).

    python tools/vdb_to_f32grid.py cloud.vdb cloud.f32grid
"""
import struct
import sys

import numpy as np


def main(src, dst):
    try:
        import pyopenvdb as vdb
    except ImportError:
        import openvdb as vdb
    grid = vdb.readAllGridMetadata(src)[0]
    grid = vdb.read(src, grid.name)                      # first grid, like openvdb::io::File::beginName (:87-88)
    lo, hi = grid.evalActiveVoxelBoundingBox()
    lo = [c - 1 for c in lo]                             # expandBy(1) (:97-101); the reference's max is exclusive,
    dims = [h - l + 2 for h, l in zip(hi, lo)]           # so the dense box is the active extent + 2 per axis
    dense = np.zeros(dims, np.float32)                   # indexed [x][y][z]
    grid.copyToArray(dense, ijk=tuple(lo))
    vol = np.ascontiguousarray(dense.transpose(2, 1, 0)) # [z][y][x], x fastest
    # the host's quantiser adds the zero border itself: hand over the payload without it
    payload = vol[1:-1, 1:-1, 1:-1]
    nz, ny, nx = payload.shape
    with open(dst, "wb") as f:
        f.write(struct.pack("<3i", nx, ny, nz))
        f.write(np.ascontiguousarray(payload).tobytes())
    print(f"{src}: first grid '{grid.name}', payload {nx}x{ny}x{nz} -> {dst}")


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    main(sys.argv[1], sys.argv[2])
