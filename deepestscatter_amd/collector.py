"""Radiance-sample producer: host-side mirror of the reference's RadianceCollector
(src/Scene/RadianceCollector.{h,cpp}) on top of ct_point_radiance_launch, plus the wire format of
its output (Persistance::Result, DeepestScatter_Train/Protocols/Result.proto).

The device part -- estimateEmission over (point, direction) tasks with per-task Welford -- runs in
libcloudtrace.so; what stays here is exactly what the reference does on the host: task
replication (scheduleTasks, :176-192), merging the replicas with PointRadianceTask::operator+=
(PointRadianceTask.h:56-68), the convergence rule (:112-118) and re-packing the unconverged tasks.
LMDB is not available on the target image, so records go to a flat file that
tools/flat_to_lmdb.py turns into the reference's LMDB layout where the `lmdb` module exists.
"""
from __future__ import annotations

import struct
from typing import Callable

import numpy as np

from .cloudtrace import POINT_TASK_DTYPE, make_point_tasks

MAX_THREAD_COUNT = 10 * 2048          # RadianceCollector.cpp:17
LAUNCHES_PER_UPDATE = 100             # :88
FLT_EPSILON = np.float32(1.1920929e-07)
f32 = np.float32


def absolute_confidence_interval(radiance, running_variance, experiment_count) -> np.float32:
    """PointRadianceTask::getAbsoluteConfidenceInterval, PointRadianceTask.h:31-36 (95 %)."""
    N = f32(experiment_count)
    sigma = np.sqrt(f32(f32(running_variance) / N))
    return f32(f32(f32(1.96) * sigma) / np.sqrt(N))


def relative_confidence_interval(radiance, running_variance, experiment_count) -> np.float32:
    """getRelativeConfidenceInterval, :23-26."""
    return f32(absolute_confidence_interval(radiance, running_variance, experiment_count) / f32(f32(radiance) + FLT_EPSILON))


def merge_tasks(into: np.void, other: np.void) -> None:
    """PointRadianceTask::operator+=, :56-68 (the reference adds the M2 values as they are)."""
    if int(other["id"]) != int(into["id"]):
        raise ValueError("Different point radiance tasks cannot be merged into one!")
    n0, n1 = int(into["experimentCount"]), int(other["experimentCount"])
    new_weight = f32(f32(f32(n1) * f32(1.0)) / f32(n0 + n1))
    into["radiance"] = f32(into["radiance"] + f32(f32(other["radiance"] - into["radiance"]) * new_weight))
    into["runningVariance"] = f32(into["runningVariance"] + other["runningVariance"])
    into["experimentCount"] = n0 + n1


class RadianceCollector:
    """init / update / isCompleted like the reference's SceneItem.  `launch(tasks, first_frame_id,
    launches)` runs the device part (default: CloudTracer.point_radiance_launch)."""

    def __init__(self, launch: Callable[[np.ndarray, int, int], np.ndarray], positions, directions,
                 batch_start_id: int = 0, max_thread_count: int = MAX_THREAD_COUNT,
                 launches_per_update: int = LAUNCHES_PER_UPDATE):
        self.launch = launch
        self.batch_start_id = batch_start_id
        self.batch_size = len(positions)
        self.max_thread_count = max_thread_count
        self.launches_per_update = launches_per_update
        self.converged_tasks: list[np.void] = []
        self.all_pixels_converged = False
        self.frame_id = 0
        # init(), :19-58: task i = (id i, point, view_direction) of record batchStartId + i
        self._schedule(make_point_tasks(positions, directions))

    # scheduleTasks, :176-192
    def _schedule(self, tasks: np.ndarray) -> None:
        self.task_repeat_count = self.max_thread_count // len(tasks)
        assert self.task_repeat_count > 0
        self.threads_count = len(tasks) * self.task_repeat_count
        buf = np.zeros(self.threads_count, POINT_TASK_DTYPE)
        r = self.task_repeat_count
        buf["id"] = np.repeat(tasks["id"], r)
        buf["position"] = np.repeat(tasks["position"], r, axis=0)
        buf["direction"] = np.repeat(tasks["direction"], r, axis=0)
        buf[0::r] = tasks                      # slot 0 keeps the accumulated statistics, replicas start fresh
        self.tasks_buffer = buf

    def get_converged_count(self) -> int:
        return len(self.converged_tasks)

    def get_remaining_count(self) -> int:
        return self.batch_size - self.get_converged_count()

    def is_completed(self) -> bool:
        return self.all_pixels_converged

    # update(), :73-141
    def update(self) -> None:
        if self.all_pixels_converged:
            return
        self.launch(self.tasks_buffer, self.frame_id + 1, self.launches_per_update)   # :88-96
        self.frame_id += self.launches_per_update
        r = self.task_repeat_count
        n = self.get_remaining_count()
        # representative += replica j, j = 1 .. r-1 (:104-108): the reference's scalar float32 operations in its order,
        # carried out for all remaining tasks at once (merge_tasks is the same arithmetic for one pair)
        buf = self.tasks_buffer[:n * r].reshape(n, r)
        rep = buf[:, 0].copy()
        rad, var, cnt = rep["radiance"].astype(f32), rep["runningVariance"].astype(f32), rep["experimentCount"].astype(np.int64)
        for j in range(1, r):
            o = buf[:, j]
            if np.any(o["id"] != rep["id"]):
                raise ValueError("Different point radiance tasks cannot be merged into one!")
            n1 = o["experimentCount"].astype(np.int64)
            new_weight = (n1.astype(f32) * f32(1.0)) / (cnt + n1).astype(f32)
            rad = rad + (o["radiance"] - rad) * new_weight
            var = var + o["runningVariance"]
            cnt = cnt + n1
        rep["radiance"], rep["runningVariance"], rep["experimentCount"] = rad, var, cnt
        with np.errstate(divide="ignore", invalid="ignore"):
            N = cnt.astype(f32)
            absolute = (f32(1.96) * np.sqrt(var / N)) / np.sqrt(N)                    # PointRadianceTask.h:31-36
            relative = absolute / (rad + FLT_EPSILON)                                 # :23-26
        converged = (relative < f32(2e-2)) | (absolute < f32(1e-4))                   # :112-114
        converged = np.where(rad < FLT_EPSILON, cnt > 100000, converged)              # :115-118
        self.converged_tasks.extend(rep[converged])
        self.all_pixels_converged = self.get_converged_count() == self.batch_size
        if not self.all_pixels_converged:
            self._schedule(rep[~converged])

    # recordToDataset(), :148-169
    def results(self) -> list[tuple[int, bytes]]:
        """(record id, serialized Persistance::Result) sorted by task id."""
        out = []
        for t in sorted(self.converged_tasks, key=lambda t: int(t["id"])):
            out.append((self.batch_start_id + int(t["id"]), encode_result(float(t["radiance"]), True)))
        return out


# ---- proto3 wire format of the three messages the path exchanges (TR/Protocols/*.proto) --------
def encode_result(light_intensity: float, is_converged: bool) -> bytes:
    """Persistance::Result { float light_intensity = 1; bool is_converged = 2; }"""
    b = b""
    if f32(light_intensity) != 0:
        b += b"\x0d" + struct.pack("<f", light_intensity)
    if is_converged:
        b += b"\x10\x01"
    return b


def decode_result(b: bytes) -> tuple[float, bool]:
    li, conv, i = 0.0, False, 0
    while i < len(b):
        tag = b[i]
        i += 1
        if tag == 0x0D:
            li = struct.unpack_from("<f", b, i)[0]
            i += 4
        elif tag == 0x10:
            conv = b[i] != 0
            i += 1
        else:
            raise ValueError(f"unexpected tag {tag:#x} in Result")
    return li, conv


def _encode_vec3(v) -> bytes:
    b = b""
    for k, x in enumerate(v):
        if f32(x) != 0:
            b += bytes([(k + 1) << 3 | 5]) + struct.pack("<f", x)
    return b


def encode_scatter_sample(scene_setup_id: int, point, view_direction) -> bytes:
    """Persistance::ScatterSample { int32 scene_setup_id = 1; Vector3 point = 2; Vector3 view_direction = 3; }"""
    b = b""
    if scene_setup_id:
        b += b"\x08" + _varint(scene_setup_id & 0xFFFFFFFFFFFFFFFF)
    for tag, v in ((0x12, point), (0x1A, view_direction)):
        body = _encode_vec3(v)
        b += bytes([tag]) + _varint(len(body)) + body
    return b


def decode_scatter_sample(b: bytes):
    sid, vecs, i = 0, {0x12: [0.0, 0.0, 0.0], 0x1A: [0.0, 0.0, 0.0]}, 0
    while i < len(b):
        tag = b[i]
        i += 1
        if tag == 0x08:
            sid, i = _read_varint(b, i)
            if sid >= 1 << 63:
                sid -= 1 << 64
        elif tag in vecs:
            n, i = _read_varint(b, i)
            j = i
            while j < i + n:
                k = b[j] >> 3
                vecs[tag][k - 1] = struct.unpack_from("<f", b, j + 1)[0]
                j += 5
            i += n
        else:
            raise ValueError(f"unexpected tag {tag:#x} in ScatterSample")
    return sid, tuple(vecs[0x12]), tuple(vecs[0x1A])


def encode_disney_descriptor(grid) -> bytes:
    """Persistance::DisneyDescriptor { bytes grid = 1; }  (DisneyDescriptor.proto:7-10): the 10*9*5*5 bytes
    DisneyDescriptorCollector::recordToDataset (DisneyDescriptorCollector.cpp:73-100) copies layer by layer."""
    raw = np.ascontiguousarray(grid, np.uint8).tobytes()
    if len(raw) != 2250:
        raise ValueError("a descriptor has 10 x 9 x 5 x 5 bytes")
    return b"\x0a" + _varint(len(raw)) + raw


def decode_disney_descriptor(b: bytes) -> np.ndarray:
    if not b:
        return np.zeros((10, 9, 5, 5), np.uint8)        # proto3 omits an empty bytes field
    if b[0] != 0x0A:
        raise ValueError(f"unexpected tag {b[0]:#x} in DisneyDescriptor")
    n, i = _read_varint(b, 1)
    return np.frombuffer(b[i:i + n], np.uint8).reshape(10, 9, 5, 5).copy()


class DisneyDescriptorCollector:
    """Host-side mirror of the reference's DisneyDescriptorCollector (src/Scene/DisneyDescriptorCollector.cpp):
    reads a batch of ScatterSample records (:25-41), runs `collect` once (:57-63) and serialises one
    Persistance::DisneyDescriptor per sample (:73-100).  `collect_fn(positions, directions)` is
    CloudTracer.collect_descriptors (ct_collect_descriptors)."""

    def __init__(self, collect_fn: Callable[[np.ndarray, np.ndarray], np.ndarray], scatter_sample_records,
                 batch_start_id: int = 0):
        self.collect_fn = collect_fn
        self.batch_start_id = batch_start_id
        samples = [decode_scatter_sample(r) for r in scatter_sample_records]
        self.positions = np.array([p for _, p, _ in samples], np.float32).reshape(-1, 3)
        self.directions = np.array([d for _, _, d in samples], np.float32).reshape(-1, 3)
        self.descriptors = None

    def collect(self) -> None:
        self.descriptors = self.collect_fn(self.positions, self.directions)

    def results(self) -> list[tuple[int, bytes]]:
        if self.descriptors is None:
            self.collect()
        return [(self.batch_start_id + i, encode_disney_descriptor(g)) for i, g in enumerate(self.descriptors)]


def _varint(n: int) -> bytes:
    out = bytearray()
    while True:
        if n < 0x80:
            out.append(n)
            return bytes(out)
        out.append((n & 0x7F) | 0x80)
        n >>= 7


def _read_varint(b: bytes, i: int):
    n, shift = 0, 0
    while True:
        n |= (b[i] & 0x7F) << shift
        i += 1
        if not b[i - 1] & 0x80:
            return n, i
        shift += 7


def write_flat_dataset(path, table: str, records: list[tuple[int, bytes]]) -> None:
    """Flat stand-in for the reference's LMDB table `table` (Dataset.h:94-98: named DB, 4-byte
    little-endian int32 keys, protobuf values): magic, table name, count, then (key, len, bytes)."""
    with open(path, "wb") as f:
        name = table.encode()
        f.write(b"DSFLAT1\0" + struct.pack("<I", len(name)) + name + struct.pack("<I", len(records)))
        for key, val in records:
            f.write(struct.pack("<iI", key, len(val)) + val)


def read_flat_dataset(path):
    b = open(path, "rb").read()
    assert b[:8] == b"DSFLAT1\0"
    n = struct.unpack_from("<I", b, 8)[0]
    table = b[12:12 + n].decode()
    count = struct.unpack_from("<I", b, 12 + n)[0]
    i, recs = 16 + n, []
    for _ in range(count):
        key, ln = struct.unpack_from("<iI", b, i)
        recs.append((key, b[i + 8:i + 8 + ln]))
        i += 8 + ln
    return table, recs
