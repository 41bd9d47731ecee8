"""fp32 build of the kernels (DWBC_F32; BASELINE config 5 names an fp32 path).  The reference computes in double only, so the
oracle stays the fp64 restatement and the tolerance is the accuracy envelope of single precision through two unpivoted
39- / 33-wide sweeps and the active-set QPs (tools/f32_accuracy.py):
    |tau_total - fp64| <= 0.1 Nm (|tau| up to ~100 Nm, i.e. 1e-3 relative) and status agreement >= 99 % on flat-contact
    configurations; tilted feet (near-singular contact blocks) are outside the envelope and are not asserted here."""
import numpy as np
import pytest

from oracle import orc
from tests import cases

TOL_F32 = 0.1  # Nm


def _oracle(q, fl, fs, tasks, lim):
    M = orc.make_model(cases.tocabi_model())
    S = orc.make_setup(cases.CONTACTS_2, tasks, lim)
    return orc.cycle_batch(M, S, q, fl, fs, 0)


@pytest.mark.parametrize("cfg", ["ds", "ss_L", "mixed"])
def test_emulated_f32_kernel_within_envelope(cfg):
    from tests.emu.emu import Emu

    B = 96
    tasks = cases.TASKS_2LEVEL
    kw = dict(seed=4242)
    if cfg == "ss_L":
        kw.update(contact_mode="L", levels=3)
        tasks = cases.TASKS_3LEVEL_SWING_R
    elif cfg == "mixed":
        kw["contact_mode"] = "mixed"
    q, fl, fs = cases.synth_batch(B, **kw)
    r = Emu(cases.URDF, cases.CONTACTS_2, tasks, cases.TAU_LIM, f32=True).run(q, fl, fs)
    tau, wr, st, _ = _oracle(q, fl, fs, tasks, cases.TAU_LIM)
    assert (r["status"] == st).mean() >= 0.97
    ok = (st == 1) & (r["status"] == 1)
    assert np.abs(r["tau"][ok].sum(axis=1) - tau[ok].sum(axis=1)).max() < TOL_F32
    assert np.abs(r["tau"][:, 0] - tau[:, 0]).max() < 0.05  # gravity torque: no QP involved


def test_emulated_f32_reduced_kernel_within_envelope():
    from tests.emu.emu import Emu
    from tests.test_reduced_path import oracle_batch

    B = 48
    q, fl, fs = cases.synth_batch(B, seed=4243)
    r = Emu(cases.URDF, cases.CONTACTS_2, cases.TASKS_2LEVEL, None, f32=True).run(q, fl, fs, reduced=True)
    tau, wr, st = oracle_batch(q, fl, fs)
    assert (r["status"] == st).mean() >= 0.9
    ok = (st == 1) & (r["status"] == 1)
    assert np.abs(r["tau"][ok][:, :2] - tau[ok][:, :2]).max() < TOL_F32  # gravity and task torque
    # the redistribution objective H = H_temp^T H_temp (dwbc.cpp:4846) is ill conditioned in single precision: looser
    assert np.median(np.abs(r["tau"][ok][:, 2] - tau[ok][:, 2]).max(axis=1)) < 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["ds", "ss_L", "mixed", "ds_yaw"])
def test_gpu_f32_kernel_within_envelope(cfg):
    """ds_yaw (random yaw, +-0.1 rad roll / pitch: tilted feet) is the distribution where single precision breaks -- the contact-null
    part of the semi-definite task QPs is decided by the 1e-8 Tikhonov weight, which fp32 cannot resolve -- so it is asserted with its own
    MEASURED envelope (profiles/r04_f32_bisect.txt, every commit from the end of round 2 to round 4 rebuilt and measured on one box:
    status agreement 0.981, median 2.2e-2, p99 0.60, 96.1 % within 0.1 Nm, max 10.8 Nm; thresholds ~1.5 x that), not with the flat-feet one."""
    import libdwbc_amd as D

    B = 1024 if cfg != "ds_yaw" else 2048
    tasks = cases.TASKS_2LEVEL
    kw = dict(seed=4244)
    if cfg == "ss_L":
        kw.update(contact_mode="L", levels=3)
        tasks = cases.TASKS_3LEVEL_SWING_R
    elif cfg == "mixed":
        kw["contact_mode"] = "mixed"
    elif cfg == "ds_yaw":
        kw.update(seed=1234, yaw=True)
    q, fl, fs = cases.synth_batch(B, **kw)
    tau, wr, st, _ = _oracle(q, fl, fs, tasks, cases.TAU_LIM)
    out = {}
    for dt in ("f32", "f64"):
        wbc = D.Batch(D.Model.from_urdf(cases.URDF), B, device=0, dtype=dt)
        for c in cases.CONTACTS_2:
            wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
        for lv, links in enumerate(tasks):
            for mode, link, pt in links:
                wbc.add_task(lv, mode, link, pt)
        wbc.set_torque_limit(np.array(cases.TAU_LIM))
        wbc.set_state(q)
        wbc.set_contact(fl)
        wbc.set_fstar_all(fs)
        wbc.solve()
        out[dt] = (wbc.get("tau"), wbc.get("wrench"), wbc.get("status"))
        if dt == "f32":
            assert wbc.kernel_name().startswith("dwbc_f32::")
    t32, w32, s32 = out["f32"]
    ok = (st == 1) & (s32 == 1)
    if cfg == "ds_yaw":
        err = np.abs(t32[ok].sum(axis=1) - tau[ok].sum(axis=1)).max(axis=1)
        assert (s32 == st).mean() >= 0.97
        assert np.median(err) < 3.5e-2 and np.quantile(err, 0.99) < 1.0 and err.max() < 20.0 and (err < TOL_F32).mean() >= 0.94
        assert np.abs(t32[ok][:, 0] - tau[ok][:, 0]).max() < 0.1  # the gravity torque (no QP) stays inside the flat-feet envelope
        assert np.abs(out["f64"][0][st == 1] - tau[st == 1]).max() < 1e-6
        return
    assert (s32 == st).mean() >= 0.99
    assert np.abs(t32[ok].sum(axis=1) - tau[ok].sum(axis=1)).max() < TOL_F32
    assert np.abs(w32[ok] - wr[ok][:, :12]).max() < 1.0  # contact wrench (N, Nm), |f_z| ~ 500 N
    # the fp64 batch next to it is untouched by the fp32 build
    assert np.abs(out["f64"][0][st == 1] - tau[st == 1]).max() < 1e-6


@pytest.mark.gpu
def test_gpu_f32_reduced_kernel_config5_within_envelope():
    """BASELINE configs[4] as written -- fp32 arithmetic x reduced (centroidal) dynamics path -- at B = 8192 on the GPU (one
    rank's share of a 65536 batch is 8192), against the fp64 numpy restatement of the reduced path on a seeded subset and
    through size-independent properties on the full batch.  Same envelope as the emulation test above."""
    import libdwbc_amd as D
    from tests.test_reduced_path import oracle_batch

    B, NS = 8192, 96
    q, fl, fs = cases.synth_batch(B, seed=4245)
    out = {}
    for dt in ("f32", "f64"):
        wbc = D.Batch(D.Model.from_urdf(cases.URDF), B, device=0, dtype=dt)
        for c in cases.CONTACTS_2:
            wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
        wbc.add_task(0, D.TASK_LINK_6D, 0)
        wbc.add_task(1, D.TASK_LINK_ROTATION, 15)
        wbc.set_state(q)
        wbc.set_contact(fl)
        wbc.set_fstar_all(fs)
        wbc.solve(reduced=True)
        out[dt] = (wbc.get("tau"), wbc.get("wrench"), wbc.get("status"))
        name = wbc.kernel_name()
        assert "dwbc_cycle_kernel_reduced" in name and name.startswith("dwbc_f32::" if dt == "f32" else "dwbc::")
    t32, w32, s32 = out["f32"]
    t64, w64, s64 = out["f64"]
    # (1) seeded subset against the oracle (numpy restatement, seconds per 100 instances)
    tau, wr, st = oracle_batch(q[:NS], fl[:NS], fs[:NS])
    ok = (st == 1) & (s32[:NS] == 1)
    both = (s32 == 1) & (s64 == 1)
    e_gt = np.abs(t32[:NS][ok][:, :2] - tau[ok][:, :2]).max(axis=(1, 2))
    e_rd = np.abs(t32[:NS][ok][:, 2] - tau[ok][:, 2]).max(axis=1)
    e_rd64 = np.abs(t32[both][:, 2] - t64[both][:, 2]).max(axis=1)
    e_tk64 = np.abs(t32[both][:, 1] - t64[both][:, 1]).max(axis=1)
    m = dict(status_vs_oracle=float((s32[:NS] == st).mean()), ok_fraction_subset=float(ok.mean()), status_vs_f64=float((s32 == s64).mean()),
             ok_fraction_batch=float(both.mean()), grav_task_max=float(e_gt.max()), redis_p50=float(np.median(e_rd)), redis_p99=float(np.percentile(e_rd, 99)),
             redis_max=float(e_rd.max()), redis_vs_f64_p50=float(np.median(e_rd64)), redis_vs_f64_p99=float(np.percentile(e_rd64, 99)),
             task_vs_f64_p99=float(np.percentile(e_tk64, 99)), task_vs_f64_max=float(e_tk64.max()),
             grav_vs_f64_max=float(np.abs(t32[:, 0] - t64[:, 0]).max()))
    print("config5 fp32 x reduced, B = 8192, measured:", {k: round(v, 6) for k, v in m.items()})
    # thresholds = what was measured on MI355X in round 3 (printed above on every run) with a margin of ~4x, not "anything goes":
    #   status_vs_oracle 1.0, ok_fraction_subset 1.0, grav_task_max 4.5e-3 Nm, redis p50 / p99 / max 1.3e-3 / 3.9e-3 / 4.4e-3 Nm,
    #   status_vs_f64 0.99988, ok_fraction_batch 0.9990, task_vs_f64_max 2.9e-3 Nm, grav_vs_f64_max 5.7e-3 Nm
    # (1) seeded subset against the fp64 restatement of the reduced path
    assert m["status_vs_oracle"] >= 0.98
    assert m["ok_fraction_subset"] > 0.95
    assert m["grav_task_max"] < 0.02          # gravity and task torque, Nm
    assert m["redis_p99"] < 0.02 and m["redis_max"] < 0.05  # redistribution torque (H_temp^T H_temp is the ill-conditioned part in fp32)
    # (2) full batch: fp32 against the fp64 kernel (itself checked against the oracle in tests/test_reduced_path.py)
    assert m["status_vs_f64"] >= 0.995
    assert m["ok_fraction_batch"] > 0.99
    assert m["grav_vs_f64_max"] < 0.02        # gravity torque: no QP involved
    assert m["task_vs_f64_max"] < 0.02 and m["redis_vs_f64_p99"] < 0.02
    assert np.isfinite(t32).all() and np.isfinite(w32).all()
    assert (w32[s32 == 1][:, 2] < 0).all() and (w32[s32 == 1][:, 8] < 0).all()  # both feet loaded (f_z < 0 convention)
