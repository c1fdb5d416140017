"""Golden digests generated with the REAL reference sources (tests/golden/make_golden_ref.py: oracle/_ref at
640x480 / 5 cm and at C4 geometry).  CPU tier: the oracle restatement reproduces every one of them bit for
bit (merged: in the reference's unordered_map bundle order).  GPU tier: the HIP path reproduces the cases
whose per-voxel update order it replays exactly (`fast` with the early-out disabled; `merged`, whose bundle
ranks in the container's order are computed on the device)."""
import os

import numpy as np
import pytest

from kimera_semantics_amd import synth
from oracle import oracle_py as O
from tests.golden.make_golden_ref import CASES, FORMS, block_digests, frame_of

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cfg(name):
    method, geom, *_rest, kw = CASES[name]
    return dict(method=0 if method == "fast" else 1, voxel_size=geom["voxel_size"], truncation_distance=geom["truncation"],
                max_ray_length_m=geom["max_ray"], semantic_measurement_probability=0.8, dynamic_labels=[20],
                label_rgba=synth.default_label_colors(), **kw)


def _check(integ, name, suffix=""):
    g = np.load(os.path.join(HERE, name + suffix + ".npz"))
    f = frame_of(name)
    assert f.xyz.shape[0] == int(g["n_points"])
    integ.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    idx = integ.block_indices()
    assert np.array_equal(idx, g["block_indices"])
    _, t, s = integ.download(idx)
    assert int((t["weight"] > 0).sum()) == int(g["touched"])
    assert np.array_equal(block_digests(t, s), g["digests"]), name


@pytest.mark.parametrize("form", sorted(FORMS))
@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_reference_golden(name, form):
    extra = dict(bundle_order=0) if CASES[name][0] == "merged" else {}
    _check(O.Oracle(O.default_config(integration_order_mode=FORMS[form][1], **_cfg(name), **extra)), name, form)


@pytest.mark.gpu
@pytest.mark.parametrize("form", sorted(FORMS))
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_reproduces_reference_golden(name, form):
    """All five cases, in both readings of the "mixed" order (`fast` with the early-out: the default mode = the serial result)."""
    from kimera_semantics_amd import binding as B
    _check(B.HipIntegrator(B.default_config(max_tiles=1 << 15, max_points=1 << 19, integration_order_mode=FORMS[form][1], **_cfg(name))), name, form)
