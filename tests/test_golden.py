"""The oracle reproduces the committed golden fixtures (tests/golden/golden_v1.npz, written by
tools/make_golden.py).  Runs on CPU; the same fixtures are checked against the HIP path in
test_gpu_parity.py::test_golden_fixture_on_gpu."""
import ctypes as C
import hashlib
from pathlib import Path

import numpy as np

import _oracle as O

GOLDEN = Path(__file__).resolve().parent / "golden" / "golden_v1.npz"


def load_golden():
    z = np.load(GOLDEN)
    out = {"volumes": {}, "inscatter": {}, "cases": [], "raw": z}
    for k in z.files:
        if k.startswith("vol_"):
            out["volumes"][k[4:]] = z[k]
        if k.startswith("ins_"):
            out["inscatter"][k[4:]] = z[k]
    for entry in z["case_names"]:
        cname, vname = str(entry).split(":")
        mode, w, h, spp = (int(v) for v in z[f"case_{cname}_meta"])
        out["cases"].append(dict(name=cname, volume=vname, mode=mode, width=w, height=h, spp=spp,
                                 mean=z[f"case_{cname}_mean"], m2=z[f"case_{cname}_m2"],
                                 counters=[int(v) for v in z[f"case_{cname}_counters"]]))
    return out


def test_rng_known_answers(oracle_lib):
    z = np.load(GOLDEN)
    for a, b, r in zip(z["tea_v0"], z["tea_v1"], z["tea_out"]):
        assert oracle_lib.orc_tea4(int(a), int(b)) == int(r)
    s = C.c_uint32(int(z["tea_out"][1]))
    assert [oracle_lib.orc_lcg(C.byref(s)) for _ in range(32)] == [int(v) for v in z["lcg_seq"]]


def test_mie_textures_digest():
    z = np.load(GOLDEN)
    tex = O.mie_textures()
    assert [hashlib.sha256(t.tobytes()).hexdigest() for t in tex] == [str(v) for v in z["mie_sha256"]]
    spot = np.array([tex[0][0], tex[0][4095], tex[1][4095], tex[2][0], tex[2][2048], tex[2][4095]], np.float32)
    assert np.array_equal(spot, z["mie_spot"])


def test_camera_golden():
    z = np.load(GOLDEN)
    assert np.array_equal(np.stack(O.camera_variables(aspect=2.0)), z["camera_uvw"])


def test_oracle_reproduces_golden_renders():
    g = load_golden()
    assert len(g["cases"]) >= 4
    for case in g["cases"]:
        for fast in (False, True):           # portable build and -mfma build agree bit for bit
            o = O.Oracle(g["volumes"][case["volume"]], case["width"], case["height"], mode=case["mode"], fast=fast)
            assert np.array_equal(o.inscatter, g["inscatter"][case["volume"]])
            mean, m2 = o.render(case["spp"])
            assert np.array_equal(mean, case["mean"]), case["name"]
            assert np.array_equal(m2, case["m2"]), case["name"]
            c = o.counters.as_dict()
            assert [c[k] for k in ("paths", "box_hits", "density_lookups", "inscatter_lookups", "scatter_events",
                                   "depth_capped")] == case["counters"]


GOLDEN_V2 = Path(__file__).resolve().parent / "golden" / "golden_v2.npz"
COUNTER_KEYS = ("paths", "box_hits", "density_lookups", "inscatter_lookups", "scatter_events", "depth_capped")


def test_oracle_reproduces_golden_v2():
    """golden_v2.npz (tools/make_golden.py): DELTA estimator, scatter-sample generator, hierarchical
    descriptors, point-radiance tasks."""
    z = np.load(GOLDEN_V2)
    for entry in z["delta_case_names"]:
        cname, vname = str(entry).split(":")
        mode, w, h, spp = (int(v) for v in z[f"case_{cname}_meta"])
        o = O.Oracle(z[f"vol_{vname}"], w, h, mode=mode, estimator=1)
        mean, m2 = o.render(spp)
        assert np.array_equal(mean, z[f"case_{cname}_mean"]) and np.array_equal(m2, z[f"case_{cname}_m2"])
        c = o.counters.as_dict()
        assert [c[k] for k in COUNTER_KEYS] == [int(v) for v in z[f"case_{cname}_counters"]]
    o = O.Oracle(z["vol_v0"], 8, 8, mode=1, cloud_size_m=700.0)
    pos, view = o.generate_scatter_samples(12, batch_seed=7)
    assert np.array_equal(pos, z["samples_pos"]) and np.array_equal(view, z["samples_dir"])
    assert np.array_equal(o.collect_descriptors(pos, view), z["descriptors"])
    from deepestscatter_amd.cloudtrace import make_point_tasks
    tasks = make_point_tasks(pos, view)
    o.point_radiance_launch(tasks, 1, 6)
    assert np.array_equal(tasks.view(np.uint8).reshape(len(tasks), 40), z["point_tasks"])
