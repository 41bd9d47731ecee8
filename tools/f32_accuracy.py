"""Accuracy envelope of the fp32 build (DWBC_F32) against the fp64 CPU restatement: torque error quantiles and status
agreement on the synthetic TOCABI batches.  `--emu` runs the kernel source compiled for the host in single precision
(tests/emu/libdwbc_emu_f32.so, no GPU needed); default runs the HIP kernels through the C-ABI."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc  # noqa: E402
from tests import cases  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--emu", action="store_true")
ap.add_argument("--batch", type=int, default=2048)
args = ap.parse_args()
B = args.batch
M = orc.make_model(cases.tocabi_model())
CFG = [("double support, flat feet", {}, cases.TASKS_2LEVEL), ("double support, random yaw + 0.1 rad tilt", dict(yaw=True), cases.TASKS_2LEVEL),
       ("left single support + swing foot", dict(contact_mode="L", levels=3), cases.TASKS_3LEVEL_SWING_R), ("mixed contact modes", dict(contact_mode="mixed"), cases.TASKS_2LEVEL)]
for name, kw, tasks in [c_ for c_ in CFG if not (os.environ.get("F32_TWO_LEVEL_ONLY") and c_[1].get("levels") == 3)]:
    q, fl, fs = cases.synth_batch(B, seed=1234, **kw)
    S = orc.make_setup(cases.CONTACTS_2, tasks, cases.TAU_LIM)
    tau, wr, st, _ = orc.cycle_batch(M, S, q, fl, fs, len(os.sched_getaffinity(0)))
    if args.emu:
        from tests.emu.emu import Emu

        r = Emu(cases.URDF, cases.CONTACTS_2, tasks, cases.TAU_LIM, f32=True).run(q, fl, fs)
        t32, s32 = r["tau"], r["status"]
    else:
        import libdwbc_amd as D

        wbc = D.Batch(D.Model.from_urdf(cases.URDF), B, device=0, dtype="f32")
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
        t32, s32 = wbc.get("tau"), wbc.get("status")
    ok = (st == 1) & (s32 == 1)
    err = np.abs(t32[ok].sum(axis=1) - tau[ok].sum(axis=1)).max(axis=1)
    eg = np.abs(t32[ok][:, 0] - tau[ok][:, 0]).max()
    print(f"{name:44s} status agreement {(s32 == st).mean():.4f} | max|tau_total - fp64| Nm: median {np.median(err):.2e} p95 {np.quantile(err, .95):.2e} "
          f"p99 {np.quantile(err, .99):.2e} max {err.max():.2e} | within 0.1 Nm {(err < 0.1).mean():.4f} | gravity torque max {eg:.1e} | |tau|max {np.abs(tau.sum(axis=1)).max():.0f}")
