"""Shared test inputs: the reference's asserted states (tests/dwbc_test.cpp:48-77,152-181,262-308 in
/root/reference) and the seeded synthetic TOCABI batches of SURVEY.md section 8d.  The set-up and the input recipe live in the
package (libdwbc_amd/workloads.py: the bench's product engine imports nothing under tests/); this module re-exports them and adds
what only the tests need (golden readers, the oracle's model, pack / URDF-variant helpers)."""
import os

import numpy as np  # noqa: F401

from libdwbc_amd.workloads import (  # noqa: F401
    CONTACTS_2, CONTACTS_4, FOOT_POINT, FSTAR_CASE, Q_CASE, TASK_LINK_6D, TASK_LINK_ROTATION, TASKS_2LEVEL, TASKS_3LEVEL_SWING_L,
    TASKS_3LEVEL_SWING_R, TAU_LIM, TOCABI_URDF, synth_batch, yaw_quat,
)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
URDF = TOCABI_URDF


def tocabi_model():
    from oracle import urdf_model

    js = os.path.join(GOLDEN, "tocabi_model.json")
    if os.path.exists(js):
        return urdf_model.model_from_json(js)
    return urdf_model.load_urdf(URDF)


def golden(case, name):
    from oracle.dwbc_np import read_golden

    return read_golden(os.path.join(GOLDEN, "cases", str(case), name))


def ensure_pack(model, tree=False):
    """the kernel pack of a model size other than TOCABI's: __graft_entry__.build() makes the ones the tests use; only a missing
    one is compiled here (two minutes of hipcc).  tree=True: the pack built for this model's own kinematic tree"""
    import libdwbc_amd as D
    from libdwbc_amd.batch import tree_tag

    name = f"libdwbc_pack_{model.ndof}_{model.nb}"
    if tree:
        name += "_t" + tree_tag([max(int(p), 0) for p in model.arrays()["parent"]])
    path = os.path.join(os.path.dirname(os.path.abspath(D.__file__)), name + ".so")
    if not os.path.exists(path):
        D.build_pack(model, tree=tree)
    return path


# ---- variants of the TOCABI fixture with joints fixed (models of other sizes for the kernel packs; RBDL merges fixed joints)
HEAD_JOINTS = ["Neck_Joint", "Head_Joint"]


def variant_urdf(path_out, fixed):
    import re

    txt = open(URDF).read()
    for j in fixed:
        txt, n = re.subn(r'(name="%s"\s+type=)"revolute"' % j, r'\1"fixed"', txt)
        assert n == 1, j
    with open(path_out, "w") as f:
        f.write(txt)
    return str(path_out)
