"""not-gpu: the C-ABI library loads and exports every symbol include/dwbc_batch.h declares; host-only entry points
(model loading, setup validation) behave like the reference; compute entry points fail loudly without a GPU."""
import os
import re

import numpy as np
import pytest

from tests import cases

ROOT = cases.ROOT


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "dwbc_batch.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dwbc_[a-z_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import ctypes

    from libdwbc_amd import _lib

    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), n
    bound = {s[0] for s in _lib.SYMBOLS}
    assert set(names) == bound, set(names) ^ bound
    _lib.load()


def test_product_urdf_reader_matches_oracle_reader():
    import libdwbc_amd as D
    from oracle import urdf_model

    m = D.Model.from_urdf(cases.URDF)
    ref = urdf_model.load_urdf(cases.URDF)
    assert (m.nb, m.ndof) == (34, 39)
    assert abs(m.total_mass - 96.211282) < 1e-9
    a = m.arrays()
    assert (a["parent"] == ref["parent"]).all()
    for k in ("R_T", "p_T", "axis", "mass", "com", "inertia"):
        assert np.abs(a[k] - np.asarray(ref[k])).max() < 1e-15, k
    # link ids used by the reference's tests (tests/dwbc_test.cpp:63-72); lookup is case-insensitive
    assert m.link_id("L_AnkleRoll_Link") == 6 and m.link_id("r_ankleroll_link") == 12
    assert m.link_id("Upperbody_Link") == 15 and m.link_id("nope") == -1
    assert m.link_name(0) == "Pelvis_Link"


def test_urdf_errors_are_reported():
    import libdwbc_amd as D

    with pytest.raises(D.DwbcError):
        D.Model.from_urdf("/nonexistent.urdf")


def test_compute_fails_loudly_without_gpu():
    import libdwbc_amd as D
    from libdwbc_amd import _lib

    if _lib.load().dwbc_device_count() > 0:
        pytest.skip("a GPU is present")
    m = D.Model.from_urdf(cases.URDF)
    with pytest.raises(D.DwbcError, match="no HIP device"):
        D.Batch(m, 4)
