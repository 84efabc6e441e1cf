"""Radiance-sample producer: host logic of RadianceCollector and the wire
format of its output.  The CPU tests drive the collector with the ORACLE as the device part; the
GPU test checks ct_point_radiance_launch against the oracle bit for bit."""
import numpy as np
import pytest

import _oracle as O
import deepestscatter_amd as ds
from deepestscatter_amd import collector as col
from conftest import sphere_volume


def sample_tasks(n, seed=0):
    rng = np.random.default_rng(seed)
    pos = (rng.random((n, 3), dtype=np.float32) - 0.5) * 0.5          # inside the cloud's box
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pos[0] = (3.0, 0.0, 0.0)
    d[0] = (1.0, 0.0, 0.0)                                              # a ray that misses the box
    pos[1] = (2.0, 0.1, 0.0)
    d[1] = (-2.0, 0.0, 0.0)                                             # enters from outside, unnormalised direction
    return pos, d


def test_result_and_scatter_sample_wire_format():
    # canonical proto3 bytes (field 1 fixed32 tag 0x0d, field 2 varint tag 0x10)
    assert col.encode_result(1.5, True) == b"\x0d\x00\x00\xc0\x3f\x10\x01"
    assert col.encode_result(0.0, False) == b""
    assert col.decode_result(col.encode_result(0.25, True)) == (0.25, True)
    try:
        from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    except Exception:
        pytest.skip("protobuf runtime not importable")
    fd = descriptor_pb2.FileDescriptorProto(name="r.proto", package="Persistance", syntax="proto3")
    m = fd.message_type.add(name="Result")
    m.field.add(name="light_intensity", number=1, type=2, label=1)
    m.field.add(name="is_converged", number=2, type=8, label=1)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    Result = message_factory.GetMessageClass(pool.FindMessageTypeByName("Persistance.Result"))
    for li, cv in [(1.5, True), (0.0, True), (3.25e-3, False)]:
        assert Result(light_intensity=li, is_converged=cv).SerializeToString() == col.encode_result(li, cv)
        r = Result.FromString(col.encode_result(li, cv))
        assert (r.light_intensity, r.is_converged) == (np.float32(li), cv)
    b = col.encode_scatter_sample(7, (0.5, 0.0, -1.25), (0.0, 1.0, 0.0))
    assert col.decode_scatter_sample(b) == (7, (0.5, 0.0, -1.25), (0.0, 1.0, 0.0))


def test_flat_dataset_roundtrip(tmp_path):
    recs = [(2048 * 3 + i, col.encode_result(0.1 * i, True)) for i in range(5)]
    col.write_flat_dataset(tmp_path / "r.flat", "Result", recs)
    assert col.read_flat_dataset(tmp_path / "r.flat") == ("Result", recs)


def test_merge_matches_the_oracle_restatement(oracle_lib):
    import ctypes as C
    rng = np.random.default_rng(2)
    for _ in range(50):
        a = np.zeros(1, ds.POINT_TASK_DTYPE)
        b = np.zeros(1, ds.POINT_TASK_DTYPE)
        a["experimentCount"], b["experimentCount"] = rng.integers(1, 5000, 2)
        a["radiance"], b["radiance"] = rng.random(2)
        a["runningVariance"], b["runningVariance"] = rng.random(2) * 10
        ref = a.copy()
        assert oracle_lib.orc_point_task_merge(ref.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)) == 0
        got = a.copy()
        col.merge_tasks(got[0], b[0])
        assert got.tobytes() == ref.tobytes()
    b["id"] = 9
    with pytest.raises(ValueError):
        col.merge_tasks(a[0], b[0])


def test_collector_logic_with_oracle_backend():
    """Replication, merge, convergence and rescheduling (RadianceCollector.cpp:73-192)."""
    tex = sphere_volume(24, seed=7)
    orc = O.Oracle(tex, 8, 8, mode=1, fast=True)
    pos, d = sample_tasks(12)
    c = col.RadianceCollector(orc.point_radiance_launch, pos, d, batch_start_id=4096, max_thread_count=48,
                              launches_per_update=25)
    assert c.task_repeat_count == 4 and c.threads_count == 48
    assert list(c.tasks_buffer["id"][:8]) == [0, 0, 0, 0, 1, 1, 1, 1]
    c.update()
    # every replica ran 25 experiments; the representative merged 4 of them
    assert c.frame_id == 25
    done = c.get_converged_count()
    total_exp = [int(t["experimentCount"]) for t in c.converged_tasks]
    assert all(e == 100 for e in total_exp)
    if not c.is_completed():
        # unconverged tasks were re-packed with more replicas, keeping their statistics in slot 0
        assert c.task_repeat_count == 48 // c.get_remaining_count()
        assert int(c.tasks_buffer[0]["experimentCount"]) == 100
        assert int(c.tasks_buffer[1]["experimentCount"]) == 0
    for _ in range(40):
        if c.is_completed():
            break
        c.update()
    assert c.get_converged_count() >= done
    # the missing ray has radiance 0: it only converges after > 1e5 experiments (:116-118)
    ids_done = {int(t["id"]) for t in c.converged_tasks}
    assert 0 not in ids_done or any(int(t["experimentCount"]) > 100000 for t in c.converged_tasks if int(t["id"]) == 0)
    recs = c.results()
    assert [k for k, _ in recs] == sorted(k for k, _ in recs) and all(k >= 4096 for k, _ in recs)
    for _, payload in recs:
        li, conv = col.decode_result(payload)
        assert conv and li >= 0


@pytest.mark.gpu
def test_point_radiance_launch_bit_exact_on_gpu():
    tex = sphere_volume(32, seed=11)
    n = 150                                                        # not a multiple of 64
    pos, d = sample_tasks(n, seed=5)
    for mode in (1, 0):
        tr = ds.CloudTracer(tex, width=8, height=8, mode=mode)
        orc = O.Oracle(tex, 8, 8, mode=mode, fast=True)
        got = tr.point_radiance_launch(ds.make_point_tasks(pos, d), 1, 7)
        ref = orc.point_radiance_launch(ds.make_point_tasks(pos, d), 1, 7)
        assert got.tobytes() == ref.tobytes()
        # a second call continues the statistics (frames 8..12), like the reference's next update()
        got = tr.point_radiance_launch(got, 8, 5)
        ref = orc.point_radiance_launch(ref, 8, 5)
        assert got.tobytes() == ref.tobytes()
        assert int(got["experimentCount"][3]) == 12 and got["radiance"][0] == 0      # task 0 misses the box
        c = tr.counters()
        o = orc.counters.as_dict()
        assert (c["density_lookups"], c["scatter_events"], c["depth_capped"]) == \
               (o["density_lookups"], o["scatter_events"], o["depth_capped"])
        tr.close()


@pytest.mark.gpu
def test_point_radiance_job_order_never_changes_a_result(monkeypatch):
    """ct_point_radiance_launch keeps its device buffers between calls and cuts a small call into single-frame jobs, handed
    out frame-major from one queue.  None of that may show in a result: calls of growing and shrinking size, re-packed task
    lists whose ids come back at other positions, against the oracle and against a handle with the round-1 job list
    (CT_POINT_ORDER=0: jobs of 8 frames in task order over 8 queues)."""
    tex = sphere_volume(32, seed=21)
    pos, d = sample_tasks(300, seed=17)
    orc = O.Oracle(tex, 8, 8, mode=1, fast=True)
    monkeypatch.setenv("CT_POINT_ORDER", "0")
    plain = ds.CloudTracer(tex, width=8, height=8, mode=1)
    monkeypatch.delenv("CT_POINT_ORDER")
    tr = ds.CloudTracer(tex, width=8, height=8, mode=1)
    rng = np.random.default_rng(3)
    tasks = ds.make_point_tasks(pos, d)
    frame = 1
    for step, (count, launches) in enumerate([(300, 3), (300, 20), (70, 9), (300, 2), (129, 33)]):
        if step == 2:
            # what a collector does between updates: the unconverged tasks, re-packed with replicas (ids repeat, move)
            keep = np.sort(rng.choice(300, 7, replace=False))
            tasks = np.repeat(tasks[keep], 10)
        elif step == 3:
            tasks = ds.make_point_tasks(pos[::-1].copy(), d[::-1].copy())            # every id now names another task
        elif step == 4:
            tasks = tasks[:129].copy()
        assert len(tasks) == count
        got = tr.point_radiance_launch(tasks.copy(), frame, launches)
        ref = orc.point_radiance_launch(tasks.copy(), frame, launches)
        old = plain.point_radiance_launch(tasks.copy(), frame, launches)
        assert got.tobytes() == ref.tobytes(), step
        assert old.tobytes() == ref.tobytes(), step
        tasks = got
        frame += launches
    assert tr.counters() == plain.counters()
    tr.close()
    plain.close()


@pytest.mark.gpu
def test_collector_end_to_end_on_gpu(tmp_path):
    tex = sphere_volume(32, seed=12)
    tr = ds.CloudTracer(tex, width=8, height=8, mode=1)             # SunMultipleScatter, Tasks.cpp:135
    pos, d = sample_tasks(40, seed=9)
    pos, d = pos[2:], d[2:]                                         # drop the contrived rays
    c = col.RadianceCollector(tr.point_radiance_launch, pos, d, batch_start_id=2048, max_thread_count=2048)
    for _ in range(30):
        if c.is_completed():
            break
        c.update()
    assert c.get_converged_count() > 0
    col.write_flat_dataset(tmp_path / "res.flat", "Result", c.results())
    table, recs = col.read_flat_dataset(tmp_path / "res.flat")
    assert table == "Result" and len(recs) == c.get_converged_count()
    tr.close()


def test_scatter_sample_generator_oracle_properties():
    """The generated points are first-scatter positions inside the cloud and the view directions
    are unit vectors (pointGeneratorCamera.cu:20-42)."""
    tex = sphere_volume(24, seed=3)
    orc = O.Oracle(tex, 8, 8, mode=1, fast=True)
    pos, d = orc.generate_scatter_samples(200, batch_seed=5)
    assert np.isfinite(pos).all() and np.isfinite(d).all()
    assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-5)
    assert np.all(np.abs(pos) <= 0.51)                              # world coordinates, box centred at 0
    # a first-scatter position sits where the density is non-zero (trilinear footprint)
    dens = [O.tex3d(tex, p + 0.5) for p in pos]
    assert min(dens) > 0
    pos2, d2 = orc.generate_scatter_samples(200, batch_seed=5)
    assert np.array_equal(pos, pos2) and np.array_equal(d, d2)
    pos3, _ = orc.generate_scatter_samples(200, batch_seed=6)
    assert not np.array_equal(pos, pos3)


@pytest.mark.gpu
def test_scatter_sample_generator_bit_exact_on_gpu():
    tex = sphere_volume(32, seed=13)
    tr = ds.CloudTracer(tex, width=8, height=8, mode=1)
    orc = O.Oracle(tex, 8, 8, mode=1, fast=True)
    for seed in (0, 77):
        gp, gd = tr.generate_scatter_samples(300, seed)
        rp, rd = orc.generate_scatter_samples(300, seed)
        assert np.array_equal(gp, rp) and np.array_equal(gd, rd)
    # generator -> collector -> records: the whole dataset path of Tasks::collect
    gp, gd = tr.generate_scatter_samples(64, 1)
    c = col.RadianceCollector(tr.point_radiance_launch, gp, gd, max_thread_count=1024, launches_per_update=50)
    c.update()
    assert c.frame_id == 50
    recs = [(i, col.encode_scatter_sample(3, gp[i], gd[i])) for i in range(len(gp))]
    assert col.decode_scatter_sample(recs[5][1])[0] == 3
    tr.close()


def test_disney_descriptor_collector_with_oracle_backend():
    """DisneyDescriptorCollector host logic + Persistance::DisneyDescriptor wire format, the oracle as device."""
    import _oracle as O
    from conftest import sphere_volume
    tex = sphere_volume(24, seed=3)
    orc = O.Oracle(tex, 8, 8, cloud_size_m=700.0)
    pos, view = orc.generate_scatter_samples(6, batch_seed=2)
    records = [col.encode_scatter_sample(4, p, d) for p, d in zip(pos, view)]
    c = col.DisneyDescriptorCollector(orc.collect_descriptors, records, batch_start_id=4 * 2048)
    out = c.results()
    assert [rid for rid, _ in out] == [4 * 2048 + i for i in range(6)]
    want = orc.collect_descriptors(pos, view)
    for (rid, blob), w in zip(out, want):
        assert len(blob) == 1 + 2 + 2250                 # tag, 2-byte varint length, payload
        assert blob[:3] == b"\x0a\xca\x11"
        assert np.array_equal(col.decode_disney_descriptor(blob), w)
    with pytest.raises(ValueError):
        col.encode_disney_descriptor(np.zeros(7, np.uint8))


def test_cpp_host_wire_format_and_dataset_equal_the_python_mirror(tmp_path):
    """host/Collectors.h (the C++ mirror of the reference's collectors) encodes Result, ScatterSample, SceneSetup and
    DisneyDescriptor records and writes its flat tables exactly like deepestscatter_amd/collector.py, whose encoders are
    checked against the protobuf runtime above: a small C++ program is compiled here and its output compared."""
    import subprocess
    from pathlib import Path
    from deepestscatter_amd import build
    root = Path(__file__).resolve().parents[1]
    build.build()
    src = tmp_path / "t.cpp"
    src.write_text(r"""
#include "Collectors.h"
using namespace DeepestScatter;
int main(int argc, char** argv)
{
    Dataset d;
    const float p[3] = { 0.5f, 0.0f, -1.25f }, v[3] = { 0.0f, 1.0f, 0.0f }, z[3] = { 0.f, 0.f, 0.f };
    d.batchAppend("ScatterSample", { Persistance::scatterSample(7, p, v), Persistance::scatterSample(0, z, z), Persistance::scatterSample(-3, v, p) }, 4096);
    d.batchAppend("Result", { Persistance::result(1.5f, true), Persistance::result(0.0f, false), Persistance::result(3.25e-3f, true) }, 10);
    std::vector<uint8_t> grid(CT_DESCRIPTOR_BYTES);
    for (size_t i = 0; i < grid.size(); i++) grid[i] = (uint8_t)(i * 7 + 3);
    d.batchAppend("DisneyDescriptor", { Persistance::disneyDescriptor(grid.data(), grid.size()) }, 2);
    const float l[3] = { -0.03f, -0.25f, 0.8f };
    d.batchAppend("SceneSetup", { Persistance::sceneSetup("clouds/a.vdb", 7000.f, l) }, 0);
    float rp[3], rv[3];
    Persistance::readScatterSample(d.getRecord("ScatterSample", 4098), rp, rv);
    if (rp[1] != 1.0f || rv[2] != -1.25f || rv[1] != 0.0f) return 3;
    CtPointRadianceTask a{}, b{};
    a.id = b.id = 5; a.experimentCount = 300; b.experimentCount = 100; a.radiance = 0.25f; b.radiance = 0.75f; a.runningVariance = 2.f; b.runningVariance = 1.f;
    RadianceCollector::merge(a, b);
    std::printf("%u %.9g %.9g %.9g %.9g\n", a.experimentCount, a.radiance, a.runningVariance, RadianceCollector::absoluteConfidenceInterval(a),
                RadianceCollector::relativeConfidenceInterval(a));
    d.save(argv[1]);
    return 0;
}
""")
    exe = tmp_path / "t"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", str(root / "deepestscatter_amd" / "host"), "-o", str(exe), str(src),
                    f"-L{root / 'deepestscatter_amd'}", "-lcloudtrace", f"-Wl,-rpath,{root / 'deepestscatter_amd'}"], check=True)
    out = tmp_path / "tables"
    r = subprocess.run([str(exe), str(out)], capture_output=True, text=True, check=True)
    recs = {t: col.read_flat_dataset(out / f"{t}.flat") for t in ("ScatterSample", "Result", "DisneyDescriptor", "SceneSetup")}
    assert recs["ScatterSample"] == ("ScatterSample", [(4096, col.encode_scatter_sample(7, (0.5, 0.0, -1.25), (0.0, 1.0, 0.0))),
                                                       (4097, col.encode_scatter_sample(0, (0, 0, 0), (0, 0, 0))),
                                                       (4098, col.encode_scatter_sample(-3, (0.0, 1.0, 0.0), (0.5, 0.0, -1.25)))])
    assert recs["Result"] == ("Result", [(10, col.encode_result(1.5, True)), (11, col.encode_result(0.0, False)), (12, col.encode_result(3.25e-3, True))])
    grid = ((np.arange(2250) * 7 + 3) & 255).astype(np.uint8)
    assert recs["DisneyDescriptor"] == ("DisneyDescriptor", [(2, col.encode_disney_descriptor(grid))])
    # SceneSetup { string cloud_path = 1; float cloud_size_m = 2; Vector3 light_direction = 3; } against the protobuf runtime
    try:
        from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
        fd = descriptor_pb2.FileDescriptorProto(name="s.proto", package="Persistance", syntax="proto3")
        v3 = fd.message_type.add(name="Vector3")
        for i, n in enumerate("xyz"):
            v3.field.add(name=n, number=i + 1, type=2, label=1)
        m = fd.message_type.add(name="SceneSetup")
        m.field.add(name="cloud_path", number=1, type=9, label=1)
        m.field.add(name="cloud_size_m", number=2, type=2, label=1)
        m.field.add(name="light_direction", number=3, type=11, label=1, type_name=".Persistance.Vector3")
        pool = descriptor_pool.DescriptorPool()
        pool.Add(fd)
        SceneSetup = message_factory.GetMessageClass(pool.FindMessageTypeByName("Persistance.SceneSetup"))
        want = SceneSetup(cloud_path="clouds/a.vdb", cloud_size_m=7000.0)
        want.light_direction.x, want.light_direction.y, want.light_direction.z = -0.03, -0.25, 0.8
        assert recs["SceneSetup"][1] == [(0, want.SerializeToString())]
    except ImportError:
        pass
    # PointRadianceTask::operator+= and the confidence intervals: the same floats as the Python mirror
    a = np.zeros(1, ds.POINT_TASK_DTYPE)[0]
    b = np.zeros(1, ds.POINT_TASK_DTYPE)[0]
    a["id"] = b["id"] = 5
    a["experimentCount"], b["experimentCount"] = 300, 100
    a["radiance"], b["radiance"], a["runningVariance"], b["runningVariance"] = 0.25, 0.75, 2.0, 1.0
    col.merge_tasks(a, b)
    got = r.stdout.split()
    assert int(got[0]) == int(a["experimentCount"]) == 400
    assert np.float32(got[1]) == a["radiance"] and np.float32(got[2]) == a["runningVariance"]
    assert np.float32(got[3]) == col.absolute_confidence_interval(a["radiance"], a["runningVariance"], a["experimentCount"])
    assert np.float32(got[4]) == col.relative_confidence_interval(a["radiance"], a["runningVariance"], a["experimentCount"])


@pytest.mark.gpu
def test_cpp_cli_collect_equals_the_python_pipeline(tmp_path):
    """`cloudtrace collect` (Tasks::collect for one scene setup: ScatterSampleCollector, RadianceCollector,
    DisneyDescriptorCollector over one batch, host/Collectors.h) writes the same tables, byte for byte, as the Python
    mirror driving the same C ABI."""
    import subprocess
    from deepestscatter_amd import build
    cli = build.build_cli()
    batch, scene_id = 96, 3
    out = tmp_path / "tables"
    r = subprocess.run([str(cli), "collect", "procedural:48", "--batch", str(batch), "--scene-id", str(scene_id), "--light", "Back",
                        "--out", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Finished writing emissions." in r.stdout and f"converged: {batch} of {batch}" in r.stdout
    tex = ds.make_procedural_cloud(48)
    tr = ds.CloudTracer(tex, width=64, height=64, mode=1, light_direction=ds.LIGHT_DIRECTIONS["Back"])   # SunMultipleScatter, Tasks.cpp:135
    start = scene_id * batch
    pos, view = tr.generate_scatter_samples(batch, start)
    samples = [(start + i, col.encode_scatter_sample(scene_id, pos[i], view[i])) for i in range(batch)]
    assert col.read_flat_dataset(out / "ScatterSample.flat") == ("ScatterSample", samples)
    rc = col.RadianceCollector(tr.point_radiance_launch, pos, view, batch_start_id=start)
    while not rc.is_completed():
        rc.update()
    assert col.read_flat_dataset(out / "Result.flat") == ("Result", rc.results())
    dc = col.DisneyDescriptorCollector(tr.collect_descriptors, [s for _, s in samples], batch_start_id=start)
    assert col.read_flat_dataset(out / "DisneyDescriptor.flat") == ("DisneyDescriptor", dc.results())
    table, setup = col.read_flat_dataset(out / "SceneSetup.flat")
    assert table == "SceneSetup" and len(setup) == 1 and setup[0][0] == scene_id and b"procedural:48" in setup[0][1]
    tr.close()


@pytest.mark.gpu
def test_cpp_cli_collect_with_scene_setups_in_flight_writes_the_same_tables(tmp_path):
    """`cloudtrace collect @list --jobs 3`: three scene setups (different clouds, lights, sizes) collected by three host threads
    with a renderer handle each, their launches overlapping on the device, write the union of the tables that three runs
    of one setup each write -- byte for byte, whatever the threads' order."""
    import subprocess
    from deepestscatter_amd import build
    cli = build.build_cli()
    rows = [("procedural:40:7", "Side", 7000.0), ("procedural:48", "Back", 5000.0), ("procedural:40:9", "-0.3,-0.8,0.2", 7000.0)]
    batch, first = 64, 5
    lst = tmp_path / "setups.txt"
    lst.write_text("# cloud light size\n" + "".join(f"{c} {l} {m}\n" for c, l, m in rows))
    out = tmp_path / "all"
    # (--gpus 0,0: the list of devices the setups are dealt to, here the one GPU twice)
    r = subprocess.run([str(cli), "collect", f"@{lst}", "--batch", str(batch), "--scene-id", str(first), "--jobs", "3", "--gpus", "0,0",
                        "--out", str(out)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("collect_timings") == 3 and '"jobs": 3' in r.stdout and '"gpus": 2' in r.stdout
    merged = {}
    for i, (c, l, m) in enumerate(rows):
        one = tmp_path / f"one{i}"
        single = tmp_path / f"setup{i}.txt"
        single.write_text(f"{c} {l} {m}\n")
        r1 = subprocess.run([str(cli), "collect", f"@{single}", "--batch", str(batch), "--scene-id", str(first + i), "--out", str(one)],
                            capture_output=True, text=True, timeout=900)
        assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-2000:]
        for table in ("SceneSetup", "ScatterSample", "Result", "DisneyDescriptor"):
            name, recs = col.read_flat_dataset(one / f"{table}.flat")
            merged.setdefault(name, []).extend(recs)
    for table, recs in merged.items():
        assert col.read_flat_dataset(out / f"{table}.flat") == (table, sorted(recs)), table
    assert len(merged["Result"]) == 3 * batch and len(merged["SceneSetup"]) == 3


def test_vectorised_update_equals_the_scalar_merge():
    """RadianceCollector.update merges the replicas of all tasks at once; the result must be what the reference's loop
    gives task by task (merge_tasks = PointRadianceTask::operator+=, the two confidence intervals, the zero-radiance rule)."""
    rng = np.random.default_rng(11)

    def fake_launch(buf, first_frame, launches):
        # any statistics will do: random radiance, variance and counts, a few all-zero tasks
        buf["radiance"] = rng.random(len(buf), dtype=np.float32) * rng.choice([0.0, 1e-3, 1.0], len(buf)).astype(np.float32)
        buf["runningVariance"] = rng.random(len(buf), dtype=np.float32) * 5
        buf["experimentCount"] = rng.integers(1, 300000, len(buf))
        return buf

    pos, d = sample_tasks(37, seed=2)
    c = col.RadianceCollector(fake_launch, pos, d, max_thread_count=37 * 6 + 5, launches_per_update=10)
    for _ in range(4):
        if c.is_completed():
            break
        r, n = c.task_repeat_count, c.get_remaining_count()
        before = len(c.converged_tasks)
        fake_launch(c.tasks_buffer, 0, 0)
        snapshot = c.tasks_buffer.copy()
        c.launch = lambda buf, a, b: buf                           # update() sees exactly `snapshot`
        c.update()
        c.launch = fake_launch
        want_conv, want_todo = [], []
        for i in range(n):
            rep = snapshot[i * r].copy()
            for j in range(1, r):
                col.merge_tasks(rep, snapshot[i * r + j])
            conv = bool(col.relative_confidence_interval(rep["radiance"], rep["runningVariance"], rep["experimentCount"]) < np.float32(2e-2)
                        or col.absolute_confidence_interval(rep["radiance"], rep["runningVariance"], rep["experimentCount"]) < np.float32(1e-4))
            if rep["radiance"] < col.FLT_EPSILON:
                conv = int(rep["experimentCount"]) > 100000
            (want_conv if conv else want_todo).append(rep)
        got_conv = c.converged_tasks[before:]
        assert [t.tobytes() for t in got_conv] == [t.tobytes() for t in want_conv]
        if want_todo:
            r2 = c.task_repeat_count
            assert [c.tasks_buffer[i * r2].tobytes() for i in range(len(want_todo))] == [t.tobytes() for t in want_todo]
