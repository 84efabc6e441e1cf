#!/usr/bin/env python3
"""Headline benchmark: Msamples/s (rays x spp) of the cloud radiance estimator on the
BASELINE.json configuration "512^3 density, 1024x1024, 1024 spp progressive" (configs[2]; the
driver's N>1 runs are configs[3]: the same job sharded by 8x8-pixel tile over N GPUs with an RCCL
reduce of the accumulated radiance buffer).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = the 1024-spp job of BASELINE.json configs[2]/[3]: --spp-per-step (default 1024) subframes of the whole
frame, whatever N is ("scaling": "strong" -- rank r renders its 1/N of the 8x8-pixel tiles for all 1024 subframes):
estimator kernel(s) + Welford accumulate kernel(s), plus (N>1) ONE RCCL SUM-reduce of the merged [mean | M2]
buffer (2 x W*H float4) to rank 0.  At N=1 a step is one launch of 1024 subframes (14 GB of per-sample scratch, which a
16 GiB slot holds at 1024^2); at N=8 one launch of 1024 subframes over an eighth of the pixels.  --weak gives every
GPU the same work per step instead (512 x N subframes per step: round 1's definition).  (The reference updates its display
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
# (multi-process GPU work on this pool: the host driver only supports dmabuf IPC; without this RCCL's peer mappings fail with
# hipIpcGetMemHandle: invalid argument.  The launcher normally exports it; a bare shell may not.)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
MEASURED_COPY_GBS = 6290.0          # the same guide: float4 copy, 79 % of spec
MEASURED_RANDOM_LINE_GBS = 7100.0   # tools/probes/gather_probe.hip: 55 G random 128-B lines per second from beyond L2


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--volume", type=int, default=512, help="density texture edge (texels)")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp-per-step", type=int, default=0,
                    help="subframes per step; default 1024 (the fixed job of BASELINE.json configs[2]/[3], strong scaling), "
                         "512 x n_gpus with --weak")
    ap.add_argument("--weak", action="store_true", help="weak scaling: 512 x n_gpus subframes per step (constant work per GPU)")
    ap.add_argument("--no-pmc-traffic", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 --pmc child runs of this command (N=1 only; "
                         "two short child runs before this process touches the GPU); fall back to profiles/pmc_latest.json")
    ap.add_argument("--mode", type=int, default=0, help="0 totalRadiance (Mie multi-scatter + NEE)")
    ap.add_argument("--estimator", type=int, default=0, choices=(0, 1),
                    help="0 MARCH = the reference's free-flight sampler (the parity path, default); "
                         "1 DELTA = Woodcock tracking over majorant cells (unbiased, not the reference's)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-delta-leg", action="store_true", help="skip the DELTA-estimator run reported beside the headline")
    ap.add_argument("--no-progressive-leg", action="store_true", help="skip the 10-subframes-per-update run (the reference's display cadence)")
    ap.add_argument("--merge", default="reduce", choices=("reduce", "gather"),
                    help="N > 1: how the shards' [mean | M2] reach rank 0 -- one SUM reduce of the full frame (default) or a gather of "
                         "every rank's own tiles, 1/N of the bytes (SURVEY section 8e)")
    ap.add_argument("--simple-kernel", action="store_true", help="A/B: one thread per pixel, nested loops")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the N>1 code path on a box with one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--group", action="store_true",
                    help="N > 1 from ONE process: the C ABI's own multi-GPU group (ct_group_*: one handle per device, ncclCommInitAll, one "
                         "ncclReduce of [mean | M2] per step) instead of one rank per GPU under torch.distributed -- the second, independent "
                         "N-GPU implementation; run it without torchrun: python bench.py --gpus N --group.  The line carries the same "
                         "frame_sha256 as the torchrun path's, so the two merged frames can be compared")
    ap.add_argument("--sync-steps", action="store_true",
                    help="wait for every step before the next one is enqueued (no path continuation across launches)")
    return ap.parse_args()


def subjob_camera(width, height, win=256):
    """The centred win x win window of the width x height frame as a frame of its own: pixel (i, j) of the small frame
    looks along the ray of pixel (x0 + i, y0 + j) of the big one when U and V shrink by win / width and win / height
    (cameraCommon.cuh:22: d = pixel / size * 2 - 1; the window is centred, so the offset term vanishes).  Same part of
    the cloud, same footprint per pixel -- the sub-job both the CPU leg and the GPU's "same sub-job" leg render."""
    import deepestscatter_amd as ds
    w, h = min(win, width), min(win, height)
    eye = (2.5, -0.4, 0.0)
    U, V, Wv = ds.calculate_camera_variables(eye, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 30.0, width / height)
    return w, h, eye, np.asarray(U, np.float32) * np.float32(w / width), np.asarray(V, np.float32) * np.float32(h / height), np.asarray(Wv, np.float32)


def cpu_baseline(ds, tex, ins, width, height, mode, target_s):
    """The oracle (our CPU port of the reference; the reference has no CPU path) on all host cores, on a bounded sample of
    the SAME workload: the centred 256x256 window (subjob_camera), and the GPU timed on that identical sub-job beside it
     -- whose frame must equal the oracle's bit for bit, so the ratio compares two computations of
    the same numbers."""
    sys.path.insert(0, str(ROOT / "tests"))
    import _oracle as O
    cores = int(O.lib(True).orc_max_threads())
    w, h, eye, U, V, Wv = subjob_camera(width, height)
    orc = O.Oracle(tex, w, h, mode=mode, fast=True, inscatter=ins, threads=cores)
    orc.set_camera(eye, U, V, Wv)
    t0 = time.perf_counter()
    orc.render_subframe(1)
    t1 = time.perf_counter() - t0
    spp = int(min(max(round(target_s / max(t1, 1e-3)), 1), 64))
    orc2 = O.Oracle(tex, w, h, mode=mode, fast=True, inscatter=ins, threads=cores)
    orc2.set_camera(eye, U, V, Wv)
    t0 = time.perf_counter()
    ref_mean, _ = orc2.render(spp)
    dt = time.perf_counter() - t0
    c = orc2.counters.as_dict()
    lookups = c["density_lookups"] + c["inscatter_lookups"]
    out = {
        "value": w * h * spp / dt / 1e6,
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": f"centred {w}x{h} window of the {width}x{height} frame as a frame of its own x {spp} spp, same volume ({dt:.1f} s)",
        "lookups_per_s": lookups / dt,
    }
    # the GPU on that identical sub-job: same camera, same subframes; two passes (the first measures the job costs)
    try:
        t = ds.CloudTracer(tex, width=w, height=h, mode=mode)
        t.set_camera(eye, U, V, Wv)
        t.render_accumulate(1, spp)
        same = bool(np.array_equal(t.mean(), ref_mean))
        t.reset()
        t0 = time.perf_counter()
        t.render_accumulate(1, spp)
        gdt = time.perf_counter() - t0
        t.close()
        out["gpu_same_subjob"] = {"value": w * h * spp / gdt / 1e6, "unit": "Msamples/s", "ms": gdt * 1e3,
                                  "bit_identical_to_the_cpu_result": same,
                                  "note": "one waited-for launch of a 65k-pixel frame: launch-bound, far below the headline rate"}
        out["gpu_over_cpu_same_subjob"] = out["gpu_same_subjob"]["value"] / out["value"]
    except Exception as e:  # the ratio is optional; the bench line is not
        out["gpu_same_subjob"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def reference_loop_leg(tr, W, H, first, spp=10, updates=32):
    """Camera::render as the reference issues it (Camera.cpp:177-230), every call waited for: isConverged(), `spp` subframes,
    the tonemap copied to the host.  What the pipelined legs are to be compared with."""
    for _ in range(2):
        tr.render_accumulate(first, spp)
        first += spp
    t0 = time.perf_counter()
    for _ in range(updates):
        tr.is_converged()
        tr.render_accumulate(first, spp)
        first += spp
        tr.tonemap(0.4)
    dt = time.perf_counter() - t0
    return {"spp_per_update": spp, "updates": updates, "ms_per_update": dt / updates * 1e3,
            "value": W * H * spp * updates / dt / 1e6, "unit": "Msamples/s"}, first


def progressive_leg(tr, W, H, first, spp=10, updates=100, ahead=0, stop=False):
    """The reference's cadence (Camera::render, Camera.cpp:189-214): `spp` subframes, then the display update.  Enqueued
    batches with ct_tonemap_async behind each; paths and unstarted jobs pass from launch to launch.
    ahead > 0: the same calls with ct_set_render_ahead(ahead) -- launches of `ahead` subframes, every call accumulating and
    displaying its own share; warm-up and timed region are whole launches, so what is timed is what is counted.
    stop: with ct_set_stop_when_converged(10, 100) -- the reference's isConverged() before every update, decided on the device
    behind every 10th subframe (this frame never passes it, so nothing freezes) -- and the host reading ct_converged_at."""
    warm = 8
    if stop:
        tr.set_stop_when_converged(10, 100)
    if ahead:
        tr.set_render_ahead(ahead)
        per = max(1, ahead // spp)
        warm, updates = 2 * per, max(per, updates // per * per)
    for _ in range(warm):                               # the ring of scratch regions and the job list for this batch size
        tr.render_accumulate_async(first, spp)
        first += spp
    tr.synchronize()
    t0 = time.perf_counter()
    for _ in range(updates):
        tr.render_accumulate_async(first, spp)
        tr.tonemap_async(0.4)
        first += spp
        if stop:
            tr.converged_at()
    tr.synchronize()
    dt = time.perf_counter() - t0
    out = {"spp_per_update": spp, "updates": updates, "tonemap_every_update": True, "ms_per_update": dt / updates * 1e3,
           "value": W * H * spp * updates / dt / 1e6, "unit": "Msamples/s"}
    if stop:
        frozen_at, tested_at, outside = tr.converged_at()
        out["convergence_test_on_device"] = {"every": 10, "frozen_at": frozen_at, "last_tested_at": tested_at, "pixels_outside_the_interval": outside}
        tr.set_stop_when_converged(0, 100)
    if ahead:
        out["render_ahead_subframes"] = ahead
        out["rendered_not_asked_for"] = tr.rendered_subframes() - (first - 1)
        tr.set_render_ahead(0)
    return out, first


PMC_PASSES = (
    # (SQ, TCC and GRBM have their own counter slots -- MI355X_MICROARCH.md "rocprofv3 PMC slots" -- so the scheduler's view rides
    # along with FETCH_SIZE; FETCH_SIZE and WRITE_SIZE need separate passes)
    ["FETCH_SIZE", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_ANY",
     "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"],
    ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"],
    # the L2's memory-side read requests by destination and size, and their in-flight level (average fabric read latency =
    # LEVEL / RDREQ): what there is on gfx950 towards "how much of the traffic is HBM" -- see hbm_split in main()
    ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_DRAM_32B_sum", "TCC_EA0_RDREQ_GMI_32B_sum", "TCC_EA0_RDREQ_LEVEL_sum"],
)


def pmc_traffic(args, S, estimator=None):
    """Hardware counters of the timed launch, measured live: three short child runs of this same command under
    `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE need separate passes: MI355X_MICROARCH.md, HBM section), started BEFORE this
    process touches the GPU.  Returns bytes across the L2's memory side for one full launch of the timed step and for one
    resume-only launch, FETCH_SIZE doubled (the guide's gfx950 correction, re-calibrated on this kernel's access
    pattern by tools/fetch_probe.py: 64 B reported per 128-B line), the scheduler's counters of the same launch (lane
    occupancy, waiting shares, instruction counts) and the memory-side request counters -- or None when the profiler cannot
    run here."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not Path(exe).exists() or "rocprof" in os.environ.get("LD_PRELOAD", "") or os.environ.get("CT_BENCH_CHILD"):
        return None
    estimator = args.estimator if estimator is None else estimator
    # (one step of the child is S subframes: one launch when its per-sample scratch fits a slot, else several equal ones)
    child = [sys.executable, str(ROOT / "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-delta-leg", "--no-progressive-leg",
             "--no-pmc-traffic", "--volume", str(args.volume), "--width", str(args.width), "--height", str(args.height),
             "--spp-per-step", str(S), "--mode", str(args.mode), "--estimator", str(estimator)]
    env = dict(os.environ, CT_BENCH_CHILD="1", TMPDIR="/tmp")
    out = {}
    tmp = tempfile.mkdtemp(prefix="ct_pmc_", dir="/tmp")
    try:
        child_line = None
        for k, counters in enumerate(PMC_PASSES):
            d = os.path.join(tmp, f"pass{k}")
            # (its own process group: on a timeout the profiler AND the child it started are ended, or the child would
            # share the GPU with the timed region)
            proc = subprocess.Popen([exe, "--pmc", *counters, "--output-format", "csv", "-d", d, "--", *child], cwd="/tmp", env=env,
                                    stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
            try:
                so, se = proc.communicate(timeout=150)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(proc.pid, signal.SIGKILL)
                proc.communicate()
                return {"error": f"rocprofv3 --pmc {' '.join(counters)} did not finish within 150 s"}
            r = subprocess.CompletedProcess(proc.args, proc.returncode, so, se)
            files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
            if r.returncode != 0 or not files:
                if k < 2:
                    return {"error": f"rocprofv3 --pmc {' '.join(counters)} failed ({r.returncode}): {(r.stderr or r.stdout)[-300:]}"}
                out["_pass2_error"] = f"rocprofv3 --pmc {' '.join(counters)} failed ({r.returncode}): {(r.stderr or r.stdout)[-200:]}"
                continue
            per = {}
            for row in csv.DictReader(open(max(files, key=os.path.getmtime))):
                if "render_persistent_kernel" in row["Kernel_Name"] or "render_delta_kernel" in row["Kernel_Name"]:
                    per.setdefault(row["Counter_Name"], []).append((int(row["Dispatch_Id"]), float(row["Counter_Value"]),
                                                                    int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
            for name, rows in per.items():
                rows.sort()
                out[name] = rows
            lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"roofline"' in l]
            if not lines:
                return {"error": "the child printed no bench line: " + (r.stderr or r.stdout)[-300:]}
            child_line = json.loads(lines[-1])
        # Launch order of the child: cost-measuring launch + rest of the warm-up step, resume-only launch of the fence,
        # the n_t launches of the timed step, resume-only launch of the final fence.  n_t from the child's own line.
        n_t = max(int(child_line["roofline"]["launches"]), 1)
        fetch, write = out["FETCH_SIZE"], out["WRITE_SIZE"]
        if len(fetch) < n_t + 1 or len(write) != len(fetch):
            return {"error": f"unexpected dispatch count {len(fetch)}/{len(write)} for {n_t} timed launches"}
        full = lambda rows: rows[-1 - n_t:-1]
        per_launch = lambda name: (sum(r[1] for r in full(out[name])) / n_t) if (name in out and len(out[name]) == len(fetch)) else None
        res = {
            "source": "rocprofv3 --pmc, three child runs of this command (1 warm-up + 1 timed step) before the timed process started: "
                      "FETCH_SIZE + the SQ counters / WRITE_SIZE TCC_HIT_sum TCC_MISS_sum / the TCC_EA0_RDREQ_* counters; traffic = FETCH_SIZE x 2 "
                      "(gfx950: a 128-B request is tallied as 64 B -- MI355X_MICROARCH.md HBM section, re-calibrated by tools/fetch_probe.py, and "
                      "confirmed by TCC_EA0_RDREQ_DRAM_32B x 32 B in the third run) + WRITE_SIZE, KiB -> bytes; Infinity-Cache hits are inside "
                      "this figure (it is traffic across the L2's memory side)",
            "full_launch_bytes": (2.0 * sum(r[1] for r in full(fetch)) + sum(r[1] for r in full(write))) * 1024.0 / n_t,
            "resume_launch_bytes": (2.0 * fetch[-1][1] + write[-1][1]) * 1024.0,
            "full_launch_ms_under_pmc": sum(r[2] for r in full(fetch)) / n_t / 1e6,
            "full_launches_measured": n_t,
            "dispatch_ms_under_pmc": [round(r[2] / 1e6, 2) for r in fetch],
        }
        miss, hit = out.get("TCC_MISS_sum"), out.get("TCC_HIT_sum")
        if miss and hit and len(miss) == len(fetch):
            res["tcc_miss_per_full_launch"] = sum(r[1] for r in full(miss)) / n_t
            res["l2_hit_rate"] = sum(r[1] for r in full(hit)) / max(sum(r[1] for r in full(hit)) + sum(r[1] for r in full(miss)), 1.0)
        # the scheduler's view of the same launch (SQ counters count quad-cycles; ratios only)
        sq = {k: per_launch(k) for k in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU",
                                         "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE")}
        if all(v for v in sq.values()):
            clock_hz = sq["GRBM_GUI_ACTIVE"] / 8.0 / (res["full_launch_ms_under_pmc"] * 1e-3)    # (summed over the 8 XCDs)
            res["sq"] = {
                "lane_occupancy": sq["SQ_THREAD_CYCLES_VALU"] / (64.0 * sq["SQ_ACTIVE_INST_VALU"]),
                "wait_any_over_wave_cycles": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
                "wait_inst_any_over_wave_cycles": sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"],
                "valu_insts_per_launch": sq["SQ_INSTS_VALU"], "salu_insts_per_launch": sq["SQ_INSTS_SALU"], "waves": sq["SQ_WAVES"],
                "clock_GHz_under_pmc": clock_hz / 1e9,
                # vector instructions issued per SIMD and cycle against one wave64 instruction per two cycles (SIMD-32)
                "valu_issue_frac_of_2_cycle_rate": 2.0 * sq["SQ_INSTS_VALU"] / 1024.0 / (sq["GRBM_GUI_ACTIVE"] / 8.0),
                "definitions": "lane_occupancy = SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU); wait_any = share of wave time "
                               "parked on s_waitcnt; wait_inst_any = share waiting to issue; clock = GRBM_GUI_ACTIVE / 8 / launch time",
            }
        ea = {k: per_launch(k) for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_DRAM_32B_sum", "TCC_EA0_RDREQ_GMI_32B_sum", "TCC_EA0_RDREQ_LEVEL_sum")}
        if ea["TCC_EA0_RDREQ_sum"]:
            res["ea"] = {
                "read_requests_per_launch": ea["TCC_EA0_RDREQ_sum"],
                "read_bytes_to_local_memory_per_launch": (ea["TCC_EA0_RDREQ_DRAM_32B_sum"] or 0.0) * 32.0,
                "read_bytes_to_other_sockets_per_launch": (ea["TCC_EA0_RDREQ_GMI_32B_sum"] or 0.0) * 32.0,
                "avg_read_latency_l2_cycles": (ea["TCC_EA0_RDREQ_LEVEL_sum"] or 0.0) / ea["TCC_EA0_RDREQ_sum"],
            }
        elif "_pass2_error" in out:
            res["ea"] = {"error": out["_pass2_error"]}
        return res
    except Exception as e:  # the figure is optional; the bench line is not
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measured_ceilings(torch, device):
    """The box's own ceilings, measured in this run ("use the measured ceiling as denominator too").
    copy: a device-to-device copy of 1 GiB (read + written bytes, best of 6) -- a streaming read+write rate, which a
    read-dominated gather that also hits the Infinity Cache can exceed.  random_line: the library's own probe
    (ct_debug_fetch_probe: one thread per 128-B line of a 4 GiB buffer, two unaligned 8-byte loads from a pseudo-random,
    never repeated line -- the estimator's access shape with every access a miss): lines x 128 B per second, from the
    difference of two calls with 12 and 4 repeats (allocation and clearing cancel out)."""
    import ctypes as C
    from deepestscatter_amd import _lib
    out = {}
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    best = None
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None or ms < best else best
    del a, b
    torch.cuda.empty_cache()
    out["copy_GBps_this_run"] = 2.0 * n * 4 / (best * 1e-3) / 1e9
    L = _lib.load()
    t = {}
    for reps in (4, 12, 4, 12):
        s_out = C.c_uint64(0)
        t0 = time.perf_counter()
        rc = L.ct_debug_fetch_probe(device, 25, reps, C.byref(s_out))
        dt = time.perf_counter() - t0
        if rc != 0:
            return out
        t[reps] = min(t.get(reps, dt), dt)
    per_repeat = (t[12] - t[4]) / 8.0
    if per_repeat > 0:
        out["random_128B_line_GBps_this_run"] = (1 << 25) * 128 / per_repeat / 1e9
    # ... and the same access shape over WORKING SETS that fit a cache level (ct_debug_fetch_probe_ws: 2^25 lanes, each one
    # pseudo-random line of the set): the line-fill ceiling of the Infinity Cache (sets up to 256 MiB) beside HBM's
    by_set = []
    for mib in (32, 64, 128, 192, 1024, 4096):
        t = {}
        for reps in (4, 12, 4, 12):
            s_out = C.c_uint64(0)
            t0 = time.perf_counter()
            if L.ct_debug_fetch_probe_ws(device, 25, mib * 8192, reps, C.byref(s_out)) != 0:
                return out
            dt = time.perf_counter() - t0
            t[reps] = min(t.get(reps, dt), dt)
        per = (t[12] - t[4]) / 8.0
        if per > 0:
            by_set.append({"working_set_MiB": mib, "lines_per_s": (1 << 25) / per, "GBps": (1 << 25) * 128 / per / 1e9})
    out["random_128B_line_by_working_set_this_run"] = by_set
    return out


def working_set_leg(ds, tex, W, H, mode, estimator, spp=64):
    """Distinct 128-B lines one launch of `spp` subframes reads (ct_debug_track_lines; a handle of its own on the diagnostics
    kernels, not timed): the kernel's working set, to be held against the 256 MiB of the Infinity Cache."""
    had = os.environ.get("CT_STATS")
    os.environ["CT_STATS"] = "1"
    try:
        t = ds.CloudTracer(tex, width=W, height=H, mode=mode, estimator=estimator)
        t.render_accumulate(1, 16)
        t.track_lines(True)
        t.render_accumulate(17, spp)
        out = t.touched_lines()
        t.close()
        out["spp_of_the_measured_launch"] = spp
        out["fits_the_256_MiB_infinity_cache"] = out["touched_MiB"] <= 256.0
        return out
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        if had is None:
            os.environ.pop("CT_STATS", None)
        else:
            os.environ["CT_STATS"] = had


def delta_leg(ds, tex, W, H, mode, S, steps, pmc=None):
    """The same workload with the DELTA estimator (Woodcock tracking, BASELINE.json north_star's algorithm; unbiased,
    not the reference's sampler, so it cannot be the parity path -- DESIGN.md 4.2), reported beside the headline:
    a second handle, one warm-up step, `steps` enqueued steps between two waits.  Not part of `value`.  `pmc`: the counters of
    this kernel's launch from pmc_traffic(..., estimator=1) -- its own roofline block."""
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
    launches = max(l1 - l0, 1)
    sec = (r1 - r0) * 1e-3
    alg = 8 * lookups + 16 * paths   # this kernel issues one fetch per counted lookup: issued bytes == algorithmic bytes
    out = {"estimator": "DELTA (Woodcock tracking over LDS-resident majorant cells)", "value": W * H * S * steps / dt / 1e6,
           "unit": "Msamples/s", "ms_per_step": dt / steps * 1e3, "kernel": "render_delta_kernel",
           "avg_launch_ms": (r1 - r0) / launches, "lookups_per_sample": lookups / max(paths, 1)}
    roof = {"bound": "hbm", "kernel": "render_delta_kernel", "peak": HBM_PEAK_GBS, "unit": "GB/s", "avg_launch_ms": (r1 - r0) / launches,
            "useful_frac": alg / sec / 1e9 / HBM_PEAK_GBS if sec > 0 else 0.0, "issued_bytes_per_launch": alg / launches}
    if pmc and "error" not in pmc:
        traffic = pmc["full_launch_bytes"] + pmc["resume_launch_bytes"] / launches
        roof.update({
            "traffic": traffic, "achieved": traffic * launches / sec / 1e9, "frac": traffic * launches / sec / 1e9 / HBM_PEAK_GBS,
            "traffic_frac": traffic * launches / sec / 1e9 / HBM_PEAK_GBS, "waste_ratio": traffic * launches / alg if alg else None,
            "l2_hit_rate": pmc.get("l2_hit_rate"), "tcc_miss_per_full_launch": pmc.get("tcc_miss_per_full_launch"),
            "full_launch_ms_under_pmc": pmc.get("full_launch_ms_under_pmc"), "sq": pmc.get("sq"), "ea": pmc.get("ea"),
            "traffic_source": pmc["source"],
            "binding_ceiling": "instruction issue, not line fills: see sq (vector issue at valu_issue_frac_of_2_cycle_rate with the scalar unit "
                               "busy beside it; waves wait to issue for wait_inst_any of their time) and DESIGN.md 4.2 'Round 4' -- 8 waves per "
                               "SIMD raise wait_inst_any from 0.30 to 0.47 and lose 6 %, and the twin-brick layout, which moves 32 % fewer bytes, "
                               "gains 1.6 %",
        })
    elif pmc:
        roof["traffic_error"] = pmc["error"]
    out["roofline"] = roof
    return out


def frame_sha256(mean, m2):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(mean).tobytes() + np.ascontiguousarray(m2).tobytes()).hexdigest()


def group_main(args):
    """bench.py --gpus N --group: the fixed 1024-spp job on N devices driven by ONE process through ct_group_* (csrc/ct_group.hip:
    one handle per device, RCCL by ncclCommInitAll, ct_group_merge = one ncclReduce of [mean | M2] per device inside a group call).
    A step = ct_group_render_accumulate (every shard's batch enqueued, all waited for) + ct_group_merge."""
    import deepestscatter_amd as ds
    N = args.gpus
    W, H = args.width, args.height
    S = args.spp_per_step if args.spp_per_step > 0 else (512 * N if args.weak else 1024)
    devices = [0] * N if args.single_device else list(range(N))
    tex = ds.make_procedural_cloud(args.volume)
    t_setup = time.perf_counter()
    g = ds.TracerGroup(tex, devices, ds.SceneParams(width=W, height=H, mode=args.mode, estimator=args.estimator))
    setup_s = time.perf_counter() - t_setup
    nxt = 1
    for _ in range(args.warmup):
        g.render_accumulate(nxt, S)
        g.merge()
        nxt += S
    t0 = time.perf_counter()
    merge_s = 0.0
    for _ in range(args.steps):
        g.render_accumulate(nxt, S)
        t1 = time.perf_counter()
        g.merge()
        merge_s += time.perf_counter() - t1
        nxt += S
    elapsed = time.perf_counter() - t0
    mean, m2 = g.mean(), g.m2()
    c = g.counters()
    out = {
        "metric": "Msamples/s (rays x spp) at 512^3 vol, 1024^2 frame; HBM GB/s vs roofline",
        "value": W * H * S * args.steps / elapsed / 1e6, "unit": "Msamples/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak" if args.weak else "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.volume}^3 procedural density, {W}x{H}, progressive, {S} spp per step (BASELINE.json configs[3]), one process, "
                               "ct_group_* below the C ABI", "volume": args.volume, "width": W, "height": H, "spp_per_step": S,
                   "parallelism": f"pixel-tile shard x{N}, ONE process: ct_group_render_accumulate + ct_group_merge (ncclCommInitAll, one ncclReduce of "
                                  "[mean | M2] per device per step" + (": devices repeated, merged by copies + an add kernel, no communicator)" if args.single_device else ")"),
                   "pipelined_steps": False},
        "multi_gpu": {"implementation": "ct_group (csrc/ct_group.hip)", "devices": devices, "merge_ms_per_step": merge_s / args.steps * 1e3},
        "roofline": None, "cpu_baseline": None, "setup_s": setup_s,
        "frame_sha256": frame_sha256(mean, m2), "subframes_in_the_frame": nxt - 1,
        "counters": c,
    }
    g.close()
    print(json.dumps(out), flush=True)


def main():
    args = parse_args()
    if args.group:
        return group_main(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        args.gpus = world

    if "rocprof" in os.environ.get("LD_PRELOAD", "") and not os.environ.get("CT_BENCH_ALL_LEGS"):
        # under a profiler (tools/gpu_pmc_*.sh, gpu_timeline.sh run this command behind rocprofv3) only the headline's launches
        # are wanted: the cadence legs' hundreds of short launches and the DELTA leg would dominate "average" and "last dispatch"
        if not (args.no_progressive_leg and args.no_delta_leg):
            print("[bench] profiler detected: skipping the progressive and DELTA legs (CT_BENCH_ALL_LEGS=1 keeps them)", file=sys.stderr)
        args.no_progressive_leg = args.no_delta_leg = True
    W, H = args.width, args.height
    S = args.spp_per_step if args.spp_per_step > 0 else (512 * world if args.weak else 1024)
    # roofline.traffic from the hardware counters, before anything here touches the GPU (child processes)
    pmc, pmc_delta = None, None
    if world == 1 and not args.no_pmc_traffic and not args.simple_kernel:
        pmc = pmc_traffic(args, S)
        if args.estimator == 0 and not args.no_delta_leg:
            pmc_delta = pmc_traffic(args, S, estimator=1)   # the DELTA leg's own roofline block

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

    t_setup = time.perf_counter()
    tex = ds.make_procedural_cloud(args.volume)
    flags = _lib.CT_FLAG_SIMPLE_KERNEL if args.simple_kernel else 0
    from deepestscatter_amd.distributed import ShardedTracer
    st = ShardedTracer(tex, ds.SceneParams(width=W, height=H, mode=args.mode, estimator=args.estimator, flags=flags), rank, world, local_rank,
                       merge=args.merge)
    tr = st.tracer
    setup_s = time.perf_counter() - t_setup

    def step(first):
        # estimator + accumulate on this rank's tiles, then (N>1) the RCCL SUM-reduce of the [mean | M2] buffer
        # (2 x W*H float4) to rank 0: tiles are disjoint, so the sum is an exact merge.  Steps are enqueued
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
    k0, f0 = tr.counters(), tr.fetch_counters()
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
    k1, f1 = tr.counters(), tr.fetch_counters()
    r1, a1, l1 = tr.kernel_time()

    own_elapsed = elapsed
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"
    t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
    multi = None
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # what the driver can check: every rank took part in the collectives on a device of its own
        per_rank = [torch.zeros(4, dtype=torch.float64, device=coll_dev) for _ in range(world)]
        merge_ms = st.merge_ms()
        mine = torch.tensor([own_elapsed / args.steps * 1e3, float(torch.cuda.current_device()), merge_ms, float(tr.kernel_time()[0] - r0) / max(args.steps, 1)],
                            dtype=torch.float64, device=coll_dev)
        dist.all_gather(per_rank, mine)
        ones = torch.ones(1, dtype=torch.float64, device=coll_dev)
        dist.all_reduce(ones)
        rows = [[float(v) for v in p.tolist()] for p in per_rank]
        multi = {
            "rccl_ranks": int(ones.item()),                     # ranks that contributed to an all-reduce of ones
            "world_size": dist.get_world_size(), "backend": dist.get_backend(),
            "devices": [int(r[1]) for r in rows],               # the device each rank rendered and reduced on
            "ms_per_step_per_rank": [r[0] for r in rows],
            "estimator_ms_per_step_per_rank": [r[3] for r in rows],
            "merge": args.merge,
            "merge_ms_per_step_per_rank": [r[2] for r in rows], # copies into the staging buffer + the collective, device time
        }
    elapsed = float(t.item())

    total_samples = W * H * S * args.steps
    value = total_samples / elapsed / 1e6

    # ---- roofline of the dominant kernel (the estimator), this rank's launches in the timed region.
    # Three byte counts per launch, all over the same HIP-event launch durations (the handle's stream):
    #   algorithmic  8 B per trilinear lookup THE ALGORITHM makes (density or shadow volume; = the oracle's counters)
    #                + the 16 B float4 result each sample writes.  The MARCH kernel does not fetch most of them (exact
    #                free-space replay, the pre-walked primary prefix, reused shadow-volume footprints), so this figure
    #                may exceed the peak: it says how much work was avoided, not how busy the memory system is.
    #   issued       8 B per footprint the kernel actually loaded (ct_fetch_counters) + 16 B per sample written.
    #   traffic      bytes across the L2's memory side from the PMC counters (every issued footprint that misses L2
    #                moves a whole 128-B line for its 16 useful bytes).
    # `achieved` = max(issued, traffic) / time -- what the memory system really moved -- and `frac` = achieved / peak.
    dk = {k: k1[k] - k0[k] for k in k1}
    df = {k: f1[k] - f0[k] for k in f1}
    launches = max(l1 - l0, 1)
    render_ms = r1 - r0
    sec = render_ms * 1e-3
    lookups = dk["density_lookups"] + dk["inscatter_lookups"]
    fetches = df["density_fetches"] + df["inscatter_fetches"]
    alg_bytes = 8 * lookups + 16 * dk["paths"]
    issued_bytes = 8 * fetches + 16 * dk["paths"]
    # (`launches` counts the full launches; the resume-only launch that ends a run of enqueued steps is booked into
    # render_ms -- it belongs to the estimator's time -- but not counted, so "per launch" means per full launch here)
    resume_launches = 0 if args.sync_steps else 1
    traffic, traffic_source, pmc_extra = None, None, {}
    if pmc and "error" not in pmc:
        traffic = (launches * pmc["full_launch_bytes"] + resume_launches * pmc["resume_launch_bytes"]) / launches
        traffic_source = pmc["source"]
        pmc_extra = {k: pmc[k] for k in ("full_launch_bytes", "resume_launch_bytes", "full_launch_ms_under_pmc", "dispatch_ms_under_pmc",
                                         "tcc_miss_per_full_launch", "l2_hit_rate", "sq", "ea") if k in pmc}
    else:
        # no profiler here: the committed figure of the same launch configuration, if there is one
        f = ROOT / "profiles" / "pmc_latest.json"
        if f.exists() and world == 1 and not args.simple_kernel and args.estimator == 0:
            try:
                rec = json.loads(f.read_text())
                lc = rec.get("launch_config", {})
                if (lc.get("volume"), lc.get("width"), lc.get("height")) == (args.volume, W, H):
                    # (the committed figure is per launch of lc.spp_per_launch subframes; traffic is proportional to the samples)
                    traffic = rec.get("hbm_bytes_per_launch") * (S * args.steps / launches) / lc.get("spp_per_launch", 512)
                    traffic_source = "profiles/pmc_latest.json (committed rocprofv3 --pmc run of this command; not measured in this run" + \
                                     (": " + pmc["error"] if pmc else "") + ")"
            except Exception:
                traffic = None
    here = {}
    if rank == 0 and world == 1:
        try:
            here = measured_ceilings(torch, local_rank)
        except Exception:
            here = {}
    achieved_bytes = max(issued_bytes / launches, traffic or 0.0)
    achieved = achieved_bytes * launches / sec / 1e9 if sec > 0 else 0.0
    # Three named fractions (round-2 review): what the kernel asked for, what the memory system moved for it, and the ratio.
    useful_GBps = issued_bytes / sec / 1e9 if sec > 0 else 0.0
    traffic_GBps = traffic * launches / sec / 1e9 if (traffic and sec > 0) else None
    # A latency view beside the bandwidth one.  Every marching lane has ONE footprint in flight (the next position needs the
    # fetched density), so Little's law bounds the line-fill rate by lanes-in-flight / miss latency; the other ceiling is the
    # rate at which the fabric fills random 128-B lines at all (the probe).  The kernel's own rate is its L2 misses per second.
    miss_per_launch = pmc_extra.get("tcc_miss_per_full_launch")
    lines_per_s = miss_per_launch * launches / sec if (miss_per_launch and sec > 0) else None
    probe_lines_per_s = (here.get("random_128B_line_GBps_this_run") or MEASURED_RANDOM_LINE_GBS) * 1e9 / 128.0
    resident_lanes = 64.0 * pmc["sq"]["waves"] if (pmc and "error" not in pmc and pmc.get("sq")) else None   # (persistent kernel: every wave resident)
    lane_occ = (pmc or {}).get("sq", {}).get("lane_occupancy") if pmc and "error" not in pmc else None   # measured in this run, or absent
    hbm_miss_ns, mall_hit_ns = 900 / 2.4, 545 / 2.4     # MI355X_MICROARCH.md, idle chip, one lane
    # the probe's rate by working set: which level do this kernel's lines come from?  (working_set: one launch's distinct lines)
    by_set = {p["working_set_MiB"]: p["lines_per_s"] for p in here.get("random_128B_line_by_working_set_this_run", [])}
    latency_model = {
        "lines_per_s": lines_per_s, "probe_random_lines_per_s": probe_lines_per_s,
        "frac_of_line_fill_rate": lines_per_s / probe_lines_per_s if lines_per_s else None,
        "probe_random_lines_per_s_by_working_set_MiB": by_set or None,
        "frac_of_line_fill_rate_of_a_128_MiB_set": lines_per_s / by_set[128] if (lines_per_s and by_set.get(128)) else None,
        "frac_of_line_fill_rate_of_a_1_GiB_set": lines_per_s / by_set[1024] if (lines_per_s and by_set.get(1024)) else None,
        "resident_lanes": resident_lanes, "lane_occupancy": lane_occ,
        "lane_occupancy_source": "SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU) of the timed launch, rocprofv3 --pmc child run of this command" if lane_occ else
                                 "not measured in this run (no profiler, N > 1 or --no-pmc-traffic)",
        "miss_latency_ns_idle_chip": {"infinity_cache_hit": mall_hit_ns, "hbm": hbm_miss_ns},
        "littles_law_ceiling_lines_per_s": resident_lanes * lane_occ / (hbm_miss_ns * 1e-9) if (resident_lanes and lane_occ) else None,
        "frac_of_littles_law_ceiling": lines_per_s / (resident_lanes * lane_occ / (hbm_miss_ns * 1e-9)) if (lines_per_s and resident_lanes and lane_occ) else None,
        "reading": "the kernel's L2 misses per second against the rate a pure random-line gather reaches over working sets of several sizes. "
                   "frac_of_line_fill_rate is against the 4 GiB probe (rounds 1-3 quoted this one), but that probe also pays for its 4 GiB of "
                   "address space: over 1 GiB -- still four times the Infinity Cache, and closer to what this kernel's brick arrays span -- the "
                   "fabric fills more lines per second, and over a set that fits the Infinity Cache more still "
                   "(probe_random_lines_per_s_by_working_set_MiB).  frac_of_line_fill_rate_of_a_1_GiB_set is the honest distance to the "
                   "line-fill ceiling.  frac_of_littles_law_ceiling: against what the resident lanes could keep in flight if they did nothing "
                   "but wait for misses (DESIGN.md 4.3)",
    }
    roofline = {
        "bound": "hbm",
        "kernel": "render_simple_kernel" if args.simple_kernel else ("render_delta_kernel" if args.estimator else "render_persistent_kernel"),
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "achieved_is": "max(issued bytes, PMC traffic) per launch / average launch duration; the traffic is bytes across the L2's memory side, "
                       "so Infinity-Cache hits are inside it: an upper bound on HBM bytes.  frac = traffic_frac; useful_frac is what the kernel asked for" if traffic else
                       "issued bytes only: no PMC traffic was measured in this run (N > 1, --no-pmc-traffic or no profiler); the "
                       "traffic-based fraction is on the N = 1 line",
        "traffic": traffic, "traffic_source": traffic_source,
        "useful_frac": useful_GBps / HBM_PEAK_GBS,        # 8 B per footprint the kernel loaded + 16 B per sample written
        "traffic_frac": traffic_GBps / HBM_PEAK_GBS if traffic_GBps else None,   # bytes across the L2's memory side (Infinity-Cache hits included)
        "waste_ratio": traffic * launches / issued_bytes if (traffic and issued_bytes) else None,   # a 128-B line per 16 requested / 8 useful bytes
        "latency_model": latency_model,
        "issued_bytes_per_launch": issued_bytes / launches,
        "issued_GBps": issued_bytes / sec / 1e9 if sec > 0 else 0.0,
        "algorithmic_bytes_per_launch": alg_bytes / launches,
        "algorithmic_GBps": alg_bytes / sec / 1e9 if sec > 0 else 0.0,
        "algorithmic_over_peak": alg_bytes / sec / 1e9 / HBM_PEAK_GBS if sec > 0 else 0.0,
        # measured ceilings ("use the measured ceiling as denominator too")
        "peak_measured": {"float4_copy_GBps": MEASURED_COPY_GBS, "random_128B_line_GBps": MEASURED_RANDOM_LINE_GBS,
                          **here,
                          "source": "MI355X_MICROARCH.md (copy); profiles/gather_probe.txt (one random 128-B line per lane, working set >= 256 MiB); "
                                    "*_this_run: measured after the timed region -- a 1 GiB device-to-device copy (read + written bytes; a "
                                    "read-dominated gather that also hits the Infinity Cache can exceed it) and the library's random-line "
                                    "probe (4 GiB, every access a miss: the ceiling for this kernel's access shape)"},
        "frac_of_measured_copy": achieved / MEASURED_COPY_GBS,
        "frac_of_random_line_probe_this_run": achieved / here["random_128B_line_GBps_this_run"] if here.get("random_128B_line_GBps_this_run") else None,
        "frac_of_random_line_probe": achieved / MEASURED_RANDOM_LINE_GBS,
        "avg_launch_ms": render_ms / launches, "launches": launches,
        "lookups_per_sample": lookups / max(dk["paths"], 1),
        "fetches_per_sample": fetches / max(dk["paths"], 1),
        "accumulate_ms_per_launch": (a1 - a0) / launches,
        "counters_per_launch": {k: v / launches for k, v in {**dk, **df}.items()},
        **pmc_extra,
    }

    if rank == 0 and world == 1 and not args.simple_kernel and pmc and "error" not in pmc and not os.environ.get("CT_BENCH_CHILD"):
        # "HBM GB/s": what the counters can and cannot say (round-3 review).  gfx950 exposes TCC, TCP, SQ, SPI, TA, TD, CPC, CPF, GRBM,
        # TCA and RDC counters to rocprofv3 (profiles/r04a/counters_list.txt: 688 names) -- no memory-controller, data-fabric or
        # Infinity-Cache block, and TCC_EA0_RDREQ_DRAM* means "local memory, as opposed to another socket or IO", which is every
        # request here (ea.read_bytes_to_local_memory_per_launch equals the FETCH_SIZE x 2 figure).  So the split cannot be measured;
        # it can be bounded: the distinct lines one launch reads (a handle on the diagnostics kernels, not timed) are what HBM must
        # deliver at least once, and whatever of that set fits the 256 MiB Infinity Cache is served on-die afterwards.
        ws = working_set_leg(ds, tex, W, H, args.mode, args.estimator)
        writes = None
        split = {"separable_by_counters": False,
                 "why": "no UMC / data-fabric / Infinity-Cache counters on gfx950 (profiles/r04a/counters_list.txt); TCC_EA0_RDREQ_DRAM* = local memory = all of it",
                 "working_set_of_one_launch": ws}
        if "error" not in ws:
            ws_bytes = (ws["density_lines_touched"] + ws["shadow_lines_touched"]) * 128.0
            split["hbm_read_bytes_per_launch_at_least"] = ws_bytes
            split["hbm_read_bytes_per_launch_at_most"] = traffic
            split["reading"] = ("the launch's reads touch %.0f MB of distinct lines: %s" % (ws_bytes / 1e6,
                                "they fit the 256 MiB Infinity Cache, so after the first touch the %.0f GB of line fills per launch are served on-die: "
                                "roofline.traffic is Infinity-Cache traffic, and HBM sees the per-sample results being written (16 B per sample) plus "
                                "one read of the working set" % (traffic / 1e9) if ws["fits_the_256_MiB_infinity_cache"] else
                                "more than the 256 MiB Infinity Cache holds, so part of the %.0f GB of line fills per launch comes from HBM -- how "
                                "much depends on how the accesses are spread over the set (the counters cannot tell); the probe's rates over 192 MiB, "
                                "1 GiB and 4 GiB sets bracket the ceiling either way" % (traffic / 1e9)))
        roofline["hbm_split"] = split
    if os.environ.get("CT_STATS"):
        roofline["scheduler_stats"] = tr.debug_stats()
    out = {
        "metric": "Msamples/s (rays x spp) at 512^3 vol, 1024^2 frame; HBM GB/s vs roofline",
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        # strong: the fixed 1024-spp job, rank r renders its 1/N of the tiles; --weak: 512 x N subframes per step
        "scaling": "weak" if args.weak else "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"{args.volume}^3 procedural density, {W}x{H}, progressive, {S} spp per step "
                        f"(BASELINE.json configs[{2 if world == 1 else 3}]: a 1024 spp job is {1024 / max(S, 1):g} such step(s)), "
                        f"mode {('totalRadiance','multipleScatterSunRadiance','singleScatterSunRadiance')[args.mode]} "
                        "(Mie multi-scatter + NEE), estimator "
                        f"{('MARCH (reference-faithful)', 'DELTA (Woodcock, LDS-resident majorant cells)')[args.estimator]}, max_depth 2000",
            "volume": args.volume, "width": W, "height": H, "spp_per_step": S,
            # the synthetic input's value distribution
            "density_stats": {"nonzero_fraction": float((tex > 0).mean()), "mean": float(tex.mean()), "max": int(tex.max()),
                              "histogram_16_bins": [int(v) for v in np.bincount(tex.reshape(-1) >> 4, minlength=16)]},
            # bytes of the volume's representations on the device (ct_debug_memory; "sparse": 0 dense march bricks, 1 row extents,
            # 2 dense addressing with sparse backing -- march_bricks_stored is then the memory behind the addresses)
            "volume_memory": tr.debug_memory(),
            "parallelism": f"pixel-tile shard x{world}" + (" + RCCL reduce of the [mean | M2] buffer per step" if world > 1 else ""),
            "pipelined_steps": not args.sync_steps,
        },
        "roofline": roofline,
        "setup_s": setup_s,
    }

    if multi:
        out["multi_gpu"] = multi
        out["rccl_ranks"] = multi["rccl_ranks"]
    if rank == 0:
        # the merged frame the timed steps produced ([mean | M2] after `subframes_in_the_frame` subframes): equal, bit for bit, to
        # the frame of `bench.py --gpus N --group` with the same --steps / --warmup, and to the N = 1 frame
        if world > 1:
            fm = st.merged.cpu().numpy()
            out["frame_sha256"] = frame_sha256(fm[0], fm[1])
        else:
            out["frame_sha256"] = frame_sha256(tr.mean(), tr.m2())
        out["subframes_in_the_frame"] = nxt - 1
    if rank == 0 and world == 1 and not args.simple_kernel and not args.no_progressive_leg and not args.sync_steps:
        prog, nxt = progressive_leg(tr, W, H, nxt)
        prog["fraction_of_headline"] = prog["value"] / value
        # the same calls served by launches of 80 subframes (ct_set_render_ahead): every update still shows the reference's
        # image for its subframe count, 80 subframes later
        ahead, nxt = progressive_leg(tr, W, H, nxt, ahead=80, stop=True)
        ahead["fraction_of_headline"] = ahead["value"] / value
        prog["with_render_ahead"] = ahead
        # ... and the reference's loop as the reference issues it: every call waited for
        waited, nxt = reference_loop_leg(tr, W, H, nxt)
        waited["fraction_of_headline"] = waited["value"] / value
        prog["reference_loop_every_call_waited_for"] = waited
        out["progressive_10spp"] = prog
    if rank == 0 and world == 1 and args.estimator == 0 and not args.simple_kernel and not args.no_delta_leg:
        out["delta_estimator"] = delta_leg(ds, tex, W, H, args.mode, S, max(args.steps, 1), pmc_delta)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ins = tr.inscatter()
        out["cpu_baseline"] = cpu_baseline(ds, tex, ins, W, H, args.mode, args.cpu_seconds)
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
