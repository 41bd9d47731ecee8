"""Rates of the paths that are not the headline: the LQP cascade and the JACC QPs on the generic HQP solver (full and reduced
model), and the fused cycle on models of other sizes (kernel packs).  Wall-clock around synchronised launches; DESIGN.md only."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libdwbc_amd as D  # noqa: E402
from tests import cases  # noqa: E402


def timed(fn, sync, reps=5):
    fn(); sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps


def make(B, tasks6=False):
    wbc = D.Batch(D.Model.from_urdf(cases.URDF), B, device=0)
    for c in cases.CONTACTS_2:
        wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    wbc.add_task(0, D.TASK_LINK_6D, 0)
    wbc.add_task(1, D.TASK_LINK_6D if tasks6 else D.TASK_LINK_ROTATION, 15)
    if not tasks6:
        wbc.set_torque_limit(np.array(cases.TAU_LIM))
    wbc.enable_dump(True)
    q, fl, fs = cases.synth_batch(B, seed=5)
    if tasks6:
        fs = np.concatenate([fs[:, :6], np.zeros((B, 3)), fs[:, 6:9]], axis=1)
    wbc.set_state(q); wbc.set_contact(fl); wbc.set_fstar_all(fs)
    return wbc


for B in (1024, 8192):
    wbc = make(B)
    wbc.solve(); wbc.sync()
    hq = D.HQP.for_lqp(wbc, 12)
    t_cfg = timed(lambda: hq.configure_lqp(wbc), wbc.sync)
    t_sol = timed(lambda: hq.solveSequential(), wbc.sync)
    print(f"B = {B:5d}  LQP (39 + 12 variables, 4 levels): configure {t_cfg * 1e3:7.2f} ms, cascade {t_sol * 1e3:8.2f} ms -> {B / (t_cfg + t_sol):9.0f} instances/s")
    hj = D.HQP.for_lqp(wbc, 12)
    t0 = timed(lambda: hj.solve_jacc(wbc, 0), wbc.sync)
    t1 = timed(lambda: hj.solve_jacc(wbc, 1), wbc.sync)
    print(f"B = {B:5d}  JACC QP level 0 / level 1: {t0 * 1e3:8.2f} / {t1 * 1e3:8.2f} ms -> {B / (t0 + t1):9.0f} instances/s (both levels)")
    wr = make(B, tasks6=True)
    wr.solve(reduced=True); wr.sync()
    hr = D.HQP.for_lqp_r(wr, 24, 12)
    t_cfg = timed(lambda: hr.configure_lqp_r(wr), wr.sync)
    t_sol = timed(lambda: hr.solveSequential(), wr.sync)
    hn = D.HQP.for_nc(wr, 21)

    def nc():
        hn.configure_lqp_r_nc(wr, hr, 1); hn.solvefirst(); hn.solveSequential()

    t_nc = timed(nc, wr.sync)
    print(f"B = {B:5d}  LQP_R (24 + 12 variables, 3 levels): configure {t_cfg * 1e3:7.2f} ms, cascade {t_sol * 1e3:8.2f} ms; LQP_R_NC (21 variables, 2 levels) {t_nc * 1e3:8.2f} ms"
          f" -> {B / (t_cfg + t_sol + t_nc):9.0f} instances/s")
    del hq, hj, hr, hn, wbc, wr

# the fused cycle on other model sizes (lean build of the pack), B = 1024
from tests.test_model_packs import VARIANTS, model_43, variant_urdf  # noqa: E402

B = 1024
for name in ("fixed_arms", "fixed_head", "plus4"):
    if name == "plus4":
        md, _ = model_43()
    else:
        md = D.Model.from_urdf(variant_urdf(f"/tmp/{name}.urdf", VARIANTS[name][0]))
    cases.ensure_pack(md)
    n = md.ndof
    wbc = D.Batch(md, B, device=0)
    for cc, l in zip(cases.CONTACTS_2, ("L_AnkleRoll_Link", "R_AnkleRoll_Link")):
        wbc.add_contact(md.link_id(l), cc["point"], cc["lx"], cc["ly"], cc["mu"], cc["muz"])
    wbc.add_task(0, D.TASK_LINK_6D, 0)
    wbc.add_task(1, D.TASK_LINK_ROTATION, md.link_id("Upperbody_Link"))
    wbc.set_torque_limit(np.full(n - 6, 300.0))
    toc = D.Model.from_urdf(cases.URDF)
    q0 = np.array(cases.Q_CASE[1])
    q = np.zeros((B, n + 1))
    q[:, :6] = q0[:6]; q[:, n] = 1.0
    for i in range(1, md.nb):
        j = toc.link_id(md.link_name(i))
        q[:, 6 + i - 1] = q0[6 + j - 1] if j > 0 else 0.1
    q[:, 6:n] += 0.01 * np.random.default_rng(1).uniform(-1, 1, size=(B, n - 6))
    fs = np.tile(np.array(list(cases.FSTAR_CASE[1][0]) + list(cases.FSTAR_CASE[1][1])), (B, 1))
    wbc.set_state(q); wbc.set_contact(np.ones((B, 2), np.uint8)); wbc.set_fstar_all(fs)
    wbc.solve(); wbc.sync()
    ms = wbc.time_solves(200) / 200
    print(f"cycle on a {n}-dof / {md.nb}-body model ({wbc.kernel_name()}): {ms * 1e3:7.1f} us per launch of {B} -> {B / ms * 1e3 / 1e6:5.2f} M cycles/s, status ok {wbc.get('status').mean():.3f}")

# the general-contact kernel (dwbc_cycle_gc.h): batches that opted into three simultaneously active contacts
for B, flags in ((1024, [1, 1, 1, 0]), (1024, [1, 1, 0, 0]), (8192, [1, 1, 1, 0])):
    wbc = D.Batch(D.Model.from_urdf(cases.URDF), B, device=0)
    for c in cases.CONTACTS_4:
        wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    wbc.add_task(0, D.TASK_LINK_6D, 0)
    wbc.add_task(1, D.TASK_LINK_ROTATION, 15)
    wbc.set_torque_limit(np.array(cases.TAU_LIM))
    wbc.set_max_active_contacts(3)
    q, _, fs = cases.synth_batch(B, seed=5, yaw=True)
    wbc.set_state(q); wbc.set_contact(np.tile(np.array(flags, np.uint8), (B, 1))); wbc.set_fstar_all(fs)
    wbc.solve(); wbc.sync()
    ms = wbc.time_solves(20) / 20
    print(f"general-contact kernel, B = {B}, flags {flags} ({wbc.kernel_name()}): {ms:8.3f} ms per launch -> {B / ms * 1e3 / 1e6:6.3f} M cycles/s, status ok {wbc.get('status').mean():.3f}")
