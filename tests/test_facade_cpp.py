"""The header-only C++ facade include/dwbc_amd.hpp (RobotData names over the C-ABI).
not-gpu: it compiles and links against libdwbc_hip.so.   gpu: the reference's CASE 1 / CASE 2 sequence
(tests/dwbc_test.cpp:29-260 in the reference) reproduces the goldens through the facade."""
import json
import os
import subprocess

import numpy as np
import pytest

from tests import cases

ROOT = cases.ROOT
EXE = os.path.join(ROOT, "tests", "cpp", "facade_case")


def _build():
    src = os.path.join(ROOT, "tests", "cpp", "facade_case.cpp")
    libdir = os.path.join(ROOT, "libdwbc_amd")
    cmd = ["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), src, "-o", EXE,
           "-L" + libdir, "-l:libdwbc_hip.so", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"]
    subprocess.check_call(cmd)


def test_facade_compiles_and_links():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [1, 2])
def test_facade_reproduces_reference_cases(case):
    _build()
    out = subprocess.check_output([EXE, cases.URDF, str(case)], text=True)
    r = json.loads(out[out.index("{"):])
    g = lambda n: cases.golden(case, n)
    e = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b).reshape(-1)).max())
    assert r["ok"] == [1, 1, 1] and r["dims"] == [39, 33, 12, 2]
    assert e(r["torque_grav_"], g("torque_grav_")) < 1e-6
    assert e(r["torque_task_"], g("torque_task_")) < 1e-6
    assert e(r["torque_contact_"], g("torque_contact_")) < (1e-8 if case == 1 else 1e-3)
    assert e(r["A_inv_"], g("A_inv_")) < 1e-8
    assert e(r["N_C"], g("N_C")) < 1e-9
    assert e(r["W"], g("W")) < 1e-8
    assert e(r["NwJw"], g("NwJw")) < 1e-9
    # before redistribution torque_contact_ = NwJw * contact_qp_(last level) (reference src/dwbc.cpp:851); the
    # redistribution adds the min-norm increment, zero here because the point is already feasible
    assert e(r["torque_contact_before_redis"], r["torque_contact_"]) < 1e-9


@pytest.mark.gpu
def test_facade_reduced_sequence():
    """The Reduced* call sequence of the reference (tests/sp_test/redu_dyn_test.cpp:263-298) through the facade: gravity and
    task torques equal the full model's on the same state (the reference's own side-by-side check, :304-317)."""
    _build()
    out = subprocess.check_output([EXE, cases.URDF, "1", "reduced"], text=True)
    r = json.loads(out[out.index("{"):])
    e = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b).reshape(-1)).max())
    assert r["ok"] == [1, 1, 1] and r["reduced_ok"] == [1, 1, 1]
    assert e(r["reduced_torque_grav_"], r["torque_grav_"]) < 1e-8
    assert e(r["reduced_torque_task_"], r["torque_task_"]) < 1e-6
    assert e(r["reduced_torque_grav_"], cases.golden(1, "torque_grav_")) < 1e-6
    assert np.abs(np.asarray(r["reduced_torque_contact_"])[12:]).max() == 0.0  # only the contact-chain joints (dwbc.cpp:3766)


@pytest.mark.gpu
def test_facade_lqp_sequence():
    """ConfigureLQP + CalcControlTorqueLQP + the torque of the answer (reference tests/sp_test/jacc_compare.cpp:386-418) through
    the facade's DWBC::HQP, against the numpy restatement on the CASE 1 state."""
    from tests.test_hqp import _lqp_oracle

    _build()
    out = subprocess.check_output([EXE, cases.URDF, "1", "lqp"], text=True)
    r = json.loads(out[out.index("{"):])
    assert r["lqp_ok"] == [1, 1, 4] and r["lqp_null"] == [45, 33, 27, 24] and r["generic"] == [33]
    q = np.array(cases.Q_CASE[1])
    fs = np.array(list(cases.FSTAR_CASE[1][0]) + list(cases.FSTAR_CASE[1][1]))
    hq, ok, tau, _ = _lqp_oracle(q, fs)
    assert ok == 1
    y = np.asarray(r["lqp_y"])
    assert (np.abs(y - hq.hqp_hs_[-1].y_ans_) / (1 + np.abs(y))).max() < 1e-6
    assert np.abs(np.asarray(r["lqp_torque"]) - tau).max() < 1e-5


@pytest.mark.gpu
def test_facade_reduced_lqp_and_jacc_sequence():
    """ConfigureLQP_R / CalcControlTorqueLQP_R, ConfigureLQP_R_NC / CalcControlTorqueLQP_R_NC, CalcSingleTaskTorqueWithJACC_QP_R
    and _R_NC (reference tests/sp_test/jacc_compare.cpp:456-487, dof_comparison_jacc.cpp:358-362) through the facade, against the
    numpy restatement on the CASE 1 state."""
    from tests.test_hqp_reduced import _oracle, _rel

    _build()
    out = subprocess.check_output([EXE, cases.URDF, "1", "lqp_r"], text=True)
    r = json.loads(out[out.index("{"):])
    assert r["lqp_r_ok"] == [1, 1, 1, 1, 3, 2] and r["jacc_r_ok"] == [1, 1] and r["reduced_dims"] == [18, 21, 12, 18, 24]
    q = np.array(cases.Q_CASE[1])
    f1 = list(cases.FSTAR_CASE[1][1])
    fs = np.array(list(cases.FSTAR_CASE[1][0]) + [0.05, -0.1, 0.02] + f1)
    o = _oracle(q, fs)
    assert _rel(np.asarray(r["lqp_r_y"]), o["lqp"].hqp_hs_[-1].y_ans_) < 2e-6
    assert _rel(np.asarray(r["lqp_r_torque"]), o["lqp_tau"]) < 1e-4
    assert _rel(np.asarray(r["lqp_nc_y"]), o["nc"].hqp_hs_[-1].y_ans_) < 1e-4
    assert _rel(np.asarray(r["jacc_r_acc"]), o["jacc"]["acc"]) < 1e-5 and _rel(np.asarray(r["jacc_r_torque"]), o["jacc"]["tau"]) < 1e-3
    assert _rel(np.asarray(r["jacc_nc_acc"]), o["jacc_nc"]["acc"]) < 1e-4
    assert np.abs(np.asarray(r["G_R"]) - o["r"].G_R).max() < 1e-8


@pytest.mark.gpu
def test_facade_rest_of_the_public_interface():
    """getContactConstraintMatrix, CalcAngularMomentumMatrix (both overloads), CopyKinematicsData into an object with another contact
    setup, ClearTaskSpace / ClearContactConstraint + a level of two links (AddTaskLink): against the numpy restatement (CASE 1 state)"""
    from oracle import dwbc_np as Dn

    _build()
    out = subprocess.check_output([EXE, cases.URDF, "1", "api"], text=True)
    r = json.loads(out[out.index("{"):out.rindex("}") + 1]) if "{" in out else {}
    q = np.array(cases.Q_CASE[1])
    m = cases.tocabi_model()
    c = Dn.Cycle(m)
    for cc in cases.CONTACTS_2:
        c.add_contact(cc["link"], cc["point"], cc["lx"], cc["ly"], cc["mu"], cc["muz"])
    c.update_kinematics(q)
    c.set_contact([1, 1])
    assert np.abs(np.asarray(r["C_contact"]).reshape(20, 12) + c.cone_matrix()).max() < 1e-12  # C = -A_const_a A_rot (dwbc.cpp:512)
    assert np.abs(np.asarray(r["cam3"]).reshape(3, 39) - c.CMM[3:]).max() < 1e-9
    cmm6 = np.asarray(r["cmm6"]).reshape(6, 39)
    assert np.abs(cmm6[:3] - c.CMM[3:]).max() < 1e-9 and np.abs(cmm6[3:] - c.CMM[:3]).max() < 1e-9
    # the copy target took over state + contact constraints + task spaces and was then switched to left single support
    c1 = Dn.Cycle(m)
    for cc in cases.CONTACTS_2:
        c1.add_contact(cc["link"], cc["point"], cc["lx"], cc["ly"], cc["mu"], cc["muz"])
    c1.update_kinematics(q)
    c1.set_contact([1, 0])
    c1.calc_contact_constraint()
    assert np.abs(np.asarray(r["rd2_torque_grav_"]) - c1.calc_grav()).max() < 1e-7
    assert abs(r["rd2_A00"][0] - c1.A[0, 0]) < 1e-9 and abs(r["rd2_A00"][1] - c1.A[7, 9]) < 1e-9 and r["rd2_A00"][2:] == [6.0, 1.0]
    # two links on level 1 (rotation of the upper body + position of the right hand)
    c2 = Dn.Cycle(m)
    for cc in cases.CONTACTS_2:
        c2.add_contact(cc["link"], cc["point"], cc["lx"], cc["ly"], cc["mu"], cc["muz"])
    c2.add_task(0, 0, 0, (0, 0, 0))
    c2.add_task(1, 6, 15, (0, 0, 0))
    c2.add_task(1, 3, m["names"].index("R_Wrist2_Link"), (0, 0, 0))
    c2.set_torque_limit(np.full(33, 300.0))
    f1 = list(cases.FSTAR_CASE[1][1])
    c2.run(q, [1, 1], [np.array(cases.FSTAR_CASE[1][0]), np.array(f1 + [0.2, -0.1, 0.3])])
    assert r["api_ok"] == [1, 2, 6] and c2.status == 1
    assert np.abs(np.asarray(r["api_torque_task_"]) - c2.tau_task).max() < 1e-6
