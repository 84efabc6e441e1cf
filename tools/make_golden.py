#!/usr/bin/env python3
"""Generate tests/golden/golden_v1.npz from the CPU oracle (oracle/ct_oracle.c).

The reference ships no golden vectors and cannot run here, so these
fixtures are produced by our restatement and pin it (and the HIP path) against drift.  Contents:
inputs (small uint8 volumes) and expected outputs (shadow volumes, running mean / M2 after N
subframes, work counters, RNG sequences, Mie texture digests, default camera frame).

    python tools/make_golden.py        # rewrites the fixture; commit the result
"""
import hashlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT))
import _oracle as O  # noqa: E402

OUT = ROOT / "tests" / "golden" / "golden_v1.npz"


def volume(dims, seed):
    nz, ny, nx = dims
    z, y, x = np.mgrid[0:nz - 2, 0:ny - 2, 0:nx - 2].astype(np.float32)
    r = np.sqrt(((x - (nx - 3) / 2) / (nx - 2)) ** 2 + ((y - (ny - 3) / 2) / (ny - 2)) ** 2 +
                ((z - (nz - 3) / 2) / (nz - 2)) ** 2)
    g = np.clip(1.0 - r / 0.45, 0, 1).astype(np.float32)
    g *= np.random.default_rng(seed).random(g.shape, dtype=np.float32)
    return O.quantize_volume(g)


def main():
    O.build(force=True)
    L = O.lib()
    data = {}
    vols = {"v0": volume((20, 20, 20), 1), "v1": volume((16, 20, 24), 2)}
    cases = [("total_v0", "v0", 0, 16, 16, 4), ("multi_v0", "v0", 1, 16, 16, 4), ("single_v0", "v0", 2, 16, 16, 4),
             ("total_v1", "v1", 0, 24, 12, 16)]
    for name, v in vols.items():
        data[f"vol_{name}"] = v
    names = []
    for cname, vname, mode, w, h, spp in cases:
        o = O.Oracle(vols[vname], w, h, mode=mode)
        mean, m2 = o.render(spp)
        c = o.counters.as_dict()
        data[f"ins_{vname}"] = o.inscatter
        data[f"case_{cname}_mean"] = mean
        data[f"case_{cname}_m2"] = m2
        data[f"case_{cname}_counters"] = np.array(
            [c[k] for k in ("paths", "box_hits", "density_lookups", "inscatter_lookups", "scatter_events",
                            "depth_capped")], np.uint64)
        data[f"case_{cname}_meta"] = np.array([mode, w, h, spp], np.int64)
        names.append(f"{cname}:{vname}")
    data["case_names"] = np.array(names)
    # RNG known answers (integer exact)
    v0 = np.array([0, 1, 4096 * 7 + 3, 0xFFFFFFFF, 123456789], np.uint64)
    v1 = np.array([0, 2, 1, 0xFFFFFFFF, 17], np.uint64)
    data["tea_v0"], data["tea_v1"] = v0, v1
    data["tea_out"] = np.array([L.orc_tea4(int(a), int(b)) for a, b in zip(v0, v1)], np.uint64)
    import ctypes as C
    s = C.c_uint32(int(data["tea_out"][1]))
    data["lcg_seq"] = np.array([L.orc_lcg(C.byref(s)) for _ in range(32)], np.uint64)
    # Mie textures
    tex = O.mie_textures()
    data["mie_sha256"] = np.array([hashlib.sha256(t.tobytes()).hexdigest() for t in tex])
    data["mie_spot"] = np.array([tex[0][0], tex[0][4095], tex[1][4095], tex[2][0], tex[2][2048], tex[2][4095]], np.float32)
    # default camera frame for the reference's 512x256 window
    U, V, W = O.camera_variables(aspect=2.0)
    data["camera_uvw"] = np.stack([U, V, W])
    OUT.parent.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT, **data)
    print("wrote", OUT, OUT.stat().st_size, "bytes")
    make_v2(vols)


def make_v2(vols):
    """golden_v2.npz: the rows added after v1 -- DELTA estimator, scatter-sample generator,
    point-radiance tasks, hierarchical descriptors (SURVEY section 8 f)."""
    out = ROOT / "tests" / "golden" / "golden_v2.npz"
    data = {"vol_v0": vols["v0"], "vol_v1": vols["v1"]}
    for cname, vname, mode, w, h, spp in (("delta_total_v0", "v0", 0, 16, 16, 4), ("delta_single_v1", "v1", 2, 24, 12, 8)):
        o = O.Oracle(vols[vname], w, h, mode=mode, estimator=1)
        mean, m2 = o.render(spp)
        c = o.counters.as_dict()
        data[f"case_{cname}_mean"], data[f"case_{cname}_m2"] = mean, m2
        data[f"case_{cname}_counters"] = np.array(
            [c[k] for k in ("paths", "box_hits", "density_lookups", "inscatter_lookups", "scatter_events",
                            "depth_capped")], np.uint64)
        data[f"case_{cname}_meta"] = np.array([mode, w, h, spp], np.int64)
    data["delta_case_names"] = np.array(["delta_total_v0:v0", "delta_single_v1:v1"])
    o = O.Oracle(vols["v0"], 8, 8, mode=1, cloud_size_m=700.0)
    pos, view = o.generate_scatter_samples(12, batch_seed=7)
    data["samples_pos"], data["samples_dir"] = pos, view
    data["descriptors"] = o.collect_descriptors(pos, view)
    from deepestscatter_amd.cloudtrace import make_point_tasks
    tasks = make_point_tasks(pos, view)
    o.point_radiance_launch(tasks, 1, 6)
    data["point_tasks"] = tasks.view(np.uint8).reshape(len(tasks), 40)
    np.savez_compressed(out, **data)
    print("wrote", out, out.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
