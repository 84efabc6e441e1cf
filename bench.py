#!/usr/bin/env python3
"""Headline benchmark: Msamples/s (rays x spp) of the cloud radiance estimator on the
BASELINE.json configuration "512^3 density, 1024x1024, 1024 spp progressive" (configs[2]; the
driver's N>1 runs are configs[3]: the same job sharded by 8x8-pixel tile over N GPUs with an RCCL
reduce of the accumulated radiance buffer).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one progressive batch of --spp-per-step (default 512 x n_gpus) subframes of the whole
frame: estimator kernel + Welford accumulate kernel, plus (N>1) the reduce of the W*H float4
radiance buffer to rank 0.  Two steps are one 1024-spp image of BASELINE.json's configuration (512 subframes
of a 1024^2 frame are what the 8 GiB per-batch sample scratch holds).  (The reference updates its display
every 10 subframes and saves every 40, Camera.cpp:189,211; a launch ends with a tail of waves that
finish its long paths unless it may hand them to the next launch, which is what the enqueued steps of this
benchmark do; measured when the kernel ran 2855 Msamples/s at 512 spp per launch: 2280 at 64, 2540 at 128, 2730
at 256; waiting for every step: 1690, 2110, 2480, 2590.)  Inputs are synthetic (procedural cloud of SURVEY.md section 8d, generated on the host
before the timed region and resident in HBM).  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--volume", type=int, default=512, help="density texture edge (texels)")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp-per-step", type=int, default=0,
                    help="subframes per progressive batch (one estimator launch + one accumulate launch); "
                         "default 512 x n_gpus, i.e. a constant number of samples per GPU per launch")
    ap.add_argument("--mode", type=int, default=0, help="0 totalRadiance (Mie multi-scatter + NEE)")
    ap.add_argument("--estimator", type=int, default=0, choices=(0, 1),
                    help="0 MARCH = the reference's free-flight sampler (the parity path, default); "
                         "1 DELTA = Woodcock tracking over majorant cells (unbiased, not the reference's)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-delta-leg", action="store_true", help="skip the DELTA-estimator run reported beside the headline")
    ap.add_argument("--simple-kernel", action="store_true", help="A/B: one thread per pixel, nested loops")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the N>1 code path on a box with one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--sync-steps", action="store_true",
                    help="wait for every step before the next one is enqueued (no path continuation across launches)")
    return ap.parse_args()


def cpu_baseline(tex, ins, width, height, mode, target_s):
    """The oracle (our CPU port of the reference; the reference has no CPU path) on all host
    cores, on a bounded sample of the SAME workload: the centred 256x256 window."""
    sys.path.insert(0, str(ROOT / "tests"))
    import _oracle as O
    cores = int(O.lib(True).orc_max_threads())
    orc = O.Oracle(tex, width, height, mode=mode, fast=True, inscatter=ins, threads=cores)
    x0, y0 = width // 2 - 128, height // 2 - 128
    win = (max(x0, 0), max(y0, 0), min(x0 + 256, width), min(y0 + 256, height))
    npix = (win[2] - win[0]) * (win[3] - win[1])
    t0 = time.perf_counter()
    orc.render_subframe(1, win)
    t1 = time.perf_counter() - t0
    spp = int(min(max(round(target_s / max(t1, 1e-3)), 1), 64))
    before = orc.counters.as_dict()
    t0 = time.perf_counter()
    for sid in range(2, 2 + spp):
        orc.render_subframe(sid, win)
    dt = time.perf_counter() - t0
    after = orc.counters.as_dict()
    lookups = (after["density_lookups"] + after["inscatter_lookups"]) - (before["density_lookups"] + before["inscatter_lookups"])
    return {
        "value": npix * spp / dt / 1e6,
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": f"centred {win[2]-win[0]}x{win[3]-win[1]} window x {spp} spp of the same volume/camera ({dt:.1f} s)",
        "lookups_per_s": lookups / dt,
    }


def delta_leg(ds, tex, W, H, mode, S, steps):
    """The same workload with the DELTA estimator (Woodcock tracking, BASELINE.json north_star's algorithm; unbiased,
    not the reference's sampler, so it cannot be the parity path -- DESIGN.md 4.2), reported beside the headline:
    a second handle, one warm-up step, `steps` enqueued steps between two waits.  Not part of `value`."""
    t = ds.CloudTracer(tex, width=W, height=H, mode=mode, estimator=1)
    t.render_accumulate_async(1, S)
    t.synchronize()
    k0, (r0, _, l0) = t.counters(), t.kernel_time()
    t0 = time.perf_counter()
    for i in range(steps):
        t.render_accumulate_async(1 + S * (i + 1), S)
    t.synchronize()
    dt = time.perf_counter() - t0
    k1, (r1, _, l1) = t.counters(), t.kernel_time()
    t.close()
    lookups = (k1["density_lookups"] - k0["density_lookups"]) + (k1["inscatter_lookups"] - k0["inscatter_lookups"])
    paths = k1["paths"] - k0["paths"]
    alg = 8 * lookups + 16 * paths
    return {"estimator": "DELTA (Woodcock tracking over LDS-resident majorant cells)", "value": W * H * S * steps / dt / 1e6,
            "unit": "Msamples/s", "ms_per_step": dt / steps * 1e3, "kernel": "render_delta_kernel",
            "avg_launch_ms": (r1 - r0) / max(l1 - l0, 1), "lookups_per_sample": lookups / max(paths, 1),
            "roofline_frac": alg / ((r1 - r0) * 1e-3) / 1e9 / HBM_PEAK_GBS if r1 > r0 else 0.0}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        args.gpus = world

    import torch
    import torch.distributed as dist

    import deepestscatter_amd as ds
    from deepestscatter_amd import _lib

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libcloudtrace has no CPU fallback)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    W, H = args.width, args.height
    S = args.spp_per_step if args.spp_per_step > 0 else 512 * world
    t_setup = time.perf_counter()
    tex = ds.make_procedural_cloud(args.volume)
    flags = _lib.CT_FLAG_SIMPLE_KERNEL if args.simple_kernel else 0
    from deepestscatter_amd.distributed import ShardedTracer
    st = ShardedTracer(tex, ds.SceneParams(width=W, height=H, mode=args.mode, estimator=args.estimator, flags=flags), rank, world, local_rank)
    tr = st.tracer
    setup_s = time.perf_counter() - t_setup

    def step(first):
        # estimator + accumulate on this rank's tiles, then (N>1) the RCCL SUM-reduce of the W*H float4
        # radiance buffer to rank 0: tiles are disjoint, so the sum is an exact merge.  Steps are enqueued
        # (ct_render_accumulate_async): a launch hands its surviving paths to the next one instead of ending
        # with a tail of waves that carry a few long paths each, so the accumulate kernel of step k (and, N>1,
        # the copy + reduce of its running mean) runs behind the launch of step k+1 and the fence at the end
        # finishes the last step with a launch that only resumes.  --sync-steps waits for every step instead
        # (every launch then runs all of its paths to their end: 2480 instead of 2680 Msamples/s).
        if args.sync_steps:
            st.step(first, S)
        else:
            st.step_async(first, S)

    def fence():
        st.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # No garbage collection from here on (a full collection takes 20-30 ms with torch imported).
    import gc
    gc.collect()
    gc.disable()
    nxt = 1
    for _ in range(args.warmup):
        step(nxt)
        nxt += S
    k0 = tr.counters()
    r0, a0, l0 = tr.kernel_time()
    fence()
    t0 = time.perf_counter()
    step_marks = []
    for _ in range(args.steps):
        step(nxt)
        nxt += S
        step_marks.append(time.perf_counter() - t0)   # host time after the call (a waited-for step has finished)
        if os.environ.get("CT_BENCH_VERBOSE") and args.sync_steps:
            step_marks.append(tr.kernel_time()[0])
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if os.environ.get("CT_BENCH_VERBOSE") and rank == 0:
        print("step end marks (ms) [and cumulative kernel ms]:", ["%.2f" % (m * 1e3 if m < 50 else m) for m in step_marks], "total %.2f" % (elapsed * 1e3), file=sys.stderr)
    k1 = tr.counters()
    r1, a1, l1 = tr.kernel_time()

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    total_samples = W * H * S * args.steps
    value = total_samples / elapsed / 1e6

    # ---- roofline of the dominant kernel (the estimator), this rank's launches in the timed region
    dk = {k: k1[k] - k0[k] for k in k1}
    launches = max(l1 - l0, 1)
    render_ms = r1 - r0
    lookups = dk["density_lookups"] + dk["inscatter_lookups"]
    # algorithmic bytes: 8 B per trilinear lookup (density or shadow volume) + the 16 B float4 result
    # each sample writes (accumulation's 64 B/pixel/batch belongs to the second, tiny kernel)
    alg_bytes = 8 * lookups + 16 * dk["paths"]
    achieved = alg_bytes / (render_ms * 1e-3) / 1e9 if render_ms > 0 else 0.0
    # HBM-side bytes per launch come from a separate rocprofv3 --pmc run of this same command (PMC
    # collection cannot run inside the timed process); profiles/pmc_latest.json holds the figure, its
    # calibration and the launch configuration it is valid for.
    traffic = None
    pmc = ROOT / "profiles" / "pmc_latest.json"
    if pmc.exists() and world == 1 and not args.simple_kernel and args.estimator == 0:
        try:
            rec = json.loads(pmc.read_text())
            lc = rec.get("launch_config", {})
            if (lc.get("spp_per_step"), lc.get("volume"), lc.get("width"), lc.get("height")) == (S, args.volume, W, H):
                traffic = rec.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {
        "bound": "hbm",
        "kernel": "render_simple_kernel" if args.simple_kernel else ("render_delta_kernel" if args.estimator else "render_persistent_kernel"),
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic,
        "algorithmic_bytes_per_launch": alg_bytes / launches,
        "avg_launch_ms": render_ms / launches, "launches": launches,
        "lookups_per_s": lookups / (render_ms * 1e-3) if render_ms > 0 else 0.0,
        "lookups_per_sample": lookups / max(dk["paths"], 1),
        "accumulate_ms_per_launch": (a1 - a0) / launches,
        "counters_per_launch": {k: v / launches for k, v in dk.items()},
    }

    if os.environ.get("CT_STATS"):
        roofline["scheduler_stats"] = tr.debug_stats()
    out = {
        "metric": "Msamples/s (rays x spp) at 512^3 vol, 1024^2 frame; HBM GB/s vs roofline",
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        # every GPU renders its 1/N of the tiles for 512 x N subframes per step: constant work per GPU per step
        "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"{args.volume}^3 procedural density, {W}x{H}, progressive, {S} spp per step "
                        f"(BASELINE.json configs[{2 if world == 1 else 3}]: a 1024 spp job is {1024 / max(S, 1):g} such steps), "
                        f"mode {('totalRadiance','multipleScatterSunRadiance','singleScatterSunRadiance')[args.mode]} "
                        "(Mie multi-scatter + NEE), estimator "
                        f"{('MARCH (reference-faithful)', 'DELTA (Woodcock, LDS-resident majorant cells)')[args.estimator]}, max_depth 2000",
            "volume": args.volume, "width": W, "height": H, "spp_per_step": S,
            "parallelism": f"pixel-tile shard x{world}" + (" + RCCL reduce of the radiance buffer" if world > 1 else ""),
            "pipelined_steps": not args.sync_steps,
        },
        "roofline": roofline,
        "setup_s": setup_s,
    }

    if rank == 0 and world == 1 and args.estimator == 0 and not args.simple_kernel and not args.no_delta_leg:
        out["delta_estimator"] = delta_leg(ds, tex, W, H, args.mode, S, max(args.steps, 1))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ins = tr.inscatter()
        out["cpu_baseline"] = cpu_baseline(tex, ins, W, H, args.mode, args.cpu_seconds)
    elif rank == 0:
        out["cpu_baseline"] = None
    tr.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
