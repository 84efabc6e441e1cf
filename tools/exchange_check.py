#!/usr/bin/env python3
"""Exchange kernels (CT_EXCHANGE=1) against the oracle and against the per-lane kernels, then an A/B timing.

    python tools/exchange_check.py [--estimator 1] [--skip-parity] [--spp 1024] [--steps 3]

Parity: small scenes, every mode, the WHOLE frame's mean / M2 / counters against the oracle twin, bit for bit, and the
watchdog count (waves that gave up on a bounded wait) must be 0.  Timing: the benchmark scene (512^3, 1024^2), waited-for
steps of --spp subframes, per-lane kernel vs exchange kernel in the same process (two handles one after the other).
"""
import argparse, json, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np

os.environ.setdefault("CT_LIBRARY", "libcloudtrace_exp.so")   # the experiments build holds these kernels (build --variant exp)
XMODE = os.environ.get("CT_EXCHANGE", "1")   # 1 = block-wide exchange, 2 = exchange within a wave


def make(tex, env, **kw):
    import deepestscatter_amd as ds
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return ds.CloudTracer(tex, **kw)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def parity(estimator):
    import _oracle as O
    import deepestscatter_amd as ds
    ok = True
    for n, size, spp, mode in [(32, 48, 8, 0), (48, 64, 6, 1), (40, 64, 8, 2), (96, 128, 24, 0)]:
        tex = ds.make_procedural_cloud(n)
        tr = make(tex, {"CT_EXCHANGE": XMODE, "CT_STATS": "1"}, width=size, height=size, mode=mode, estimator=estimator)
        tr.render_accumulate(1, spp)            # first 32 subframes of a pose: the cost-measuring launch (per-lane kernel)
        tr.render_accumulate(spp + 1, spp)      # exchange kernel
        tr.render_accumulate(2 * spp + 1, 40)   # exchange kernel, long enough to cycle the rings
        total = 2 * spp + 40
        mean, m2, c, st = tr.mean(), tr.m2(), tr.counters(), tr.debug_stats()
        orc = O.Oracle(tex, size, size, mode=mode, fast=True, estimator=estimator, inscatter=tr.inscatter())
        rm, rm2 = orc.render(total)
        good = np.array_equal(mean, rm) and np.array_equal(m2, rm2) and c == orc.counters.as_dict() and st["watchdog"] == 0
        print(f"parity n={n} size={size} spp={total} mode={mode}: {'OK' if good else 'MISMATCH'} watchdog={st['watchdog']} "
              f"tracking lanes/visit {st['raw'][3] / max(st['raw'][2], 1):.1f} scatter lanes/batch {st['raw'][5] / max(st['raw'][4], 1):.1f}", flush=True)
        if not good:
            print("   counters", c, orc.counters.as_dict(), "diff pixels", int((mean != rm).any(-1).sum()))
        ok = ok and good
        tr.close()
    return ok


def timing(estimator, spp, steps, volume, size, stats=False):
    import deepestscatter_amd as ds
    tex = ds.make_procedural_cloud(volume)
    out = {}
    variants = [("per_lane", {"CT_EXCHANGE": "0", "CT_CONTINUATION": "0"}), ("exchange", {"CT_EXCHANGE": XMODE})]
    if stats:
        variants = [("exchange_stats", {"CT_EXCHANGE": XMODE, "CT_STATS": "1"})]
    for name, env in variants:
        tr = make(tex, env, width=size, height=size, estimator=estimator)
        tr.render_accumulate(1, 32)
        first = 33
        tr.render_accumulate(first, spp); first += spp          # warm-up (scratch, job list)
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.render_accumulate(first, spp); first += spp
        dt = (time.perf_counter() - t0) / steps
        rms, ams, launches = tr.kernel_time()
        out[name] = {"ms_per_step": dt * 1e3, "Msamples_per_s": size * size * spp / dt / 1e6, "mean_checksum": float(tr.mean().astype(np.float64).sum())}
        print(name, json.dumps(out[name]), flush=True)
        if stats:
            r = tr.debug_stats()["raw"]
            print("stats: regen phases %d lanes/phase %.1f | tracking visits %d lanes/visit %.1f | scatter batches %d lanes/batch %.1f | adoptions %d lanes/adoption %.1f | idle spins %d | watchdog %d"
                  % (r[0], r[1] / max(r[0], 1), r[2], r[3] / max(r[2], 1), r[4], r[5] / max(r[4], 1), r[33], r[34] / max(r[33], 1), r[35], r[63]), flush=True)
        tr.close()
    if stats:
        return out
    assert out["per_lane"]["mean_checksum"] == out["exchange"]["mean_checksum"], "the two kernels disagree"
    print("exchange_speedup", out["exchange"]["Msamples_per_s"] / out["per_lane"]["Msamples_per_s"])
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--estimator", type=int, default=1)
    ap.add_argument("--skip-parity", action="store_true")
    ap.add_argument("--skip-timing", action="store_true")
    ap.add_argument("--spp", type=int, default=512)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--volume", type=int, default=512)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--stats", action="store_true", help="timing leg: only the exchange kernel's diagnostics build, print its phase statistics")
    a = ap.parse_args()
    if not a.skip_parity and not parity(a.estimator):
        sys.exit(1)
    if not a.skip_timing:
        timing(a.estimator, a.spp, a.steps, a.volume, a.size, a.stats)
