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


EXE_G = os.path.join(ROOT, "tests", "cpp", "facade_general")
Q_DC = [-0.0325, -0.0579, 0.7273, 0.0194, -0.0118, -0.0008, -0.0006, 0.0698, -0.7835, 1.6487, -0.8420, -0.0911,
        -0.0007, 0.0767, -0.7963, 1.6742, -0.8549, -0.1150, -0.0001, -0.0003, 0.0204,
        0.2998, 0.3001, 1.5000, -1.2701, -1.0507, 0.0000, -1.0000, 0.0000, -0.0000, 0.0003,
        -0.2998, -0.3060, -1.5001, 1.2700, 1.0848, 0.0000, 1.0000, 0.0000, 0.9997]


def _build_general():
    src = os.path.join(ROOT, "tests", "cpp", "facade_general.cpp")
    libdir = os.path.join(ROOT, "libdwbc_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), src, "-o", EXE_G, "-L" + libdir, "-l:libdwbc_hip.so",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])


def test_facade_general_sequence_compiles():
    _build_general()
    assert os.path.exists(EXE_G)


@pytest.mark.gpu
def test_facade_both_hands_on_one_level_and_a_third_contact():
    """`AddTaskSpace(3, ...)` twice (a 12-dof level, reference tests/sp_test/data_confirmation.cpp:66-70) with
    the harness's warm repetitions (:91-107), and `SetContact(true, true, true)` -- through the drop-in class, against the C restatement
    on the same inputs."""
    from oracle import orc

    _build_general()
    out = subprocess.check_output([EXE_G, cases.URDF], text=True)
    r = json.loads(out[out.index("{"):])
    e = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b).reshape(-1)).max())
    M = orc.make_model(cases.tocabi_model())
    con = [dict(link=l, point=cases.FOOT_POINT, lx=lx, ly=ly, mu=0.2, muz=0.2) for l, lx, ly in ((6, 0.12, 0.06), (12, 0.12, 0.06), (23, 0.04, 0.04), (31, 0.04, 0.04))]
    q = np.array([Q_DC])
    q[0, [3, 4, 5, 39]] /= np.linalg.norm(q[0, [3, 4, 5, 39]])  # (four-digit quaternion of the harness: unit length here and in the C++ program)
    # both hands on the last level
    P, R, T6 = 3, cases.TASK_LINK_ROTATION, cases.TASK_LINK_6D  # TASK_LINK_POSITION = 3 (include/dwbc_task.h:23-33)
    tasks = [[(P, 0, (0, 0, 0))], [(R, 15, (0, 0, 0))], [(R, 25, (0, 0, 0))], [(T6, 23, (0, 0, 0)), (T6, 33, (0, 0, 0))]]
    fs = np.array([[0.3142, -1.8202, -1.7750, -1.78677, 0.84977, 0.10850, -0.85340, 0.85992, 0.12655,
                    0.40251, 0.39975, 0.75672, -0.82841, 3.03652, 0.08954, 0.27585, 0.37898, 0.93234, -0.95724, 4.38036, 0.25202]])
    S = orc.make_setup(con, tasks, cases.TAU_LIM)
    tau, wr, st, _ = orc.cycle_batch(M, S, q, np.array([[1, 1, 0, 0]], np.uint8), fs, 1)
    assert st[0] == 1 and r["ok"] == [1, 1, 1] and r["dims"] == [4, 12, 12]
    assert e(r["torque_grav_"], tau[0, 0]) < 1e-6 and e(r["torque_task_"], tau[0, 1]) < 1e-6 and e(r["torque_contact_"], tau[0, 2]) < 1e-6
    assert e(r["contact_force"], wr[0, :12]) < 1e-5
    assert r["warm_ok"] == 6 and e(r["warm_torque_task_"], tau[0, 1]) < 1e-6
    # a third contact
    S3 = orc.make_setup(con, cases.TASKS_2LEVEL, cases.TAU_LIM)
    tau3, wr3, st3, _ = orc.cycle_batch(M, S3, q, np.array([[1, 1, 1, 0]], np.uint8), np.array([[0.1, 0.4, 0.1, 0.1, -0.1, 0.1, 0.1, -0.1, 0.1]]), 1)
    assert st3[0] == 1 and r["c3_ok"] == [1, 1, 1, 18]
    assert e(r["c3_torque_grav_"], tau3[0, 0]) < 1e-6 and e(r["c3_torque_task_"], tau3[0, 1]) < 1e-6 and e(r["c3_torque_contact_"], tau3[0, 2]) < 1e-6
    assert e(r["c3_contact_force"], wr3[0, :18]) < 1e-5
