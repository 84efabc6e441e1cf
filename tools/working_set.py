#!/usr/bin/env python3
"""The estimators' WORKING SET on the benchmark scene (round 4; ct_debug_track_lines, diagnostics kernels): distinct 128-B
lines of the density array and of the shadow volume's bricks that launches of S subframes read, against the 256 MiB of the
Infinity Cache.  A set that fits is served on-die after its first touch: `roofline.traffic` (bytes across the L2's memory
side) is then Infinity-Cache traffic, not HBM traffic.

    CT_STATS=1 python tools/working_set.py [--volume 512] [--size 1024] [--spp 16 64 256 1024] [--estimator 0 1] [--nee 1 2]
"""
import argparse, json, os, sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--volume", type=int, default=512)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--spp", type=int, nargs="+", default=[16, 64, 256, 1024])
    ap.add_argument("--estimator", type=int, nargs="+", default=[0, 1])
    a = ap.parse_args()
    os.environ.setdefault("CT_STATS", "1")
    import deepestscatter_amd as ds
    tex = ds.make_procedural_cloud(a.volume)
    out = []
    for est in a.estimator:
        tr = ds.CloudTracer(tex, width=a.size, height=a.size, estimator=est)
        tr.render_accumulate(1, 16)              # (the cost-measuring launch of the pose)
        tr.track_lines(True)
        first = 17
        for S in a.spp:
            tr.render_accumulate(first, S)
            first += S
            t = tr.touched_lines(clear=True)
            t.update({"estimator": ("MARCH", "DELTA")[est], "spp_per_launch": S, "delta_nee": os.environ.get("CT_DELTA_NEE", "default") if est else None,
                      "fraction_of_density_array": t["density_lines_touched"] / max(t["density_lines"], 1),
                      "fraction_of_shadow_array": t["shadow_lines_touched"] / max(t["shadow_lines"], 1),
                      "fits_the_256_MiB_infinity_cache": t["touched_MiB"] <= 256})
            out.append(t)
            print(json.dumps(t), flush=True)
        tr.close()


if __name__ == "__main__":
    main()
