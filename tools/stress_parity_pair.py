"""GPU parity sweep of the two-wave kernel (dwbc_cycle2p.h: batches of at most 4 instances per CU, B = 1024 here) against
oracle/dwbc_oracle.c: torques, wrench, status on seeded two-level batches (flat, tilted, mixed contact modes, no torque limit)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import libdwbc_amd as D  # noqa: E402
from oracle import orc  # noqa: E402
from tests import cases  # noqa: E402

M = orc.make_model(cases.tocabi_model())
NS = int(os.environ.get("STRESS_SEEDS", "8"))
B = 1024
cfgs = {"ds": ({}, cases.TAU_LIM), "ds_yaw": (dict(yaw=True), cases.TAU_LIM), "mixed": (dict(contact_mode="mixed"), cases.TAU_LIM),
        "mixed_yaw": (dict(contact_mode="mixed", yaw=True), cases.TAU_LIM), "ds_nolim": ({}, None)}
for name, (kw, lim) in cfgs.items():
    w = D.Batch(D.Model.from_urdf(cases.URDF), B, device=0)
    for c in cases.CONTACTS_2:
        w.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    for lv, links in enumerate(cases.TASKS_2LEVEL):
        for mode, link, pt in links:
            w.add_task(lv, mode, link, pt)
    w.set_torque_limit(None if lim is None else np.array(lim))
    S = orc.make_setup(cases.CONTACTS_2, cases.TASKS_2LEVEL, lim)
    worst = worstw = 0.0
    mism = tot = okc = 0
    for seed in range(NS):
        q, fl, fs = cases.synth_batch(B, seed=9300 + seed, **kw)
        w.set_state(q); w.set_contact(fl); w.set_fstar_all(fs); w.solve()
        tau, wrn, st = w.get("tau"), w.get("wrench"), w.get("status")
        tr, wr, sr, _ = orc.cycle_batch(M, S, q, fl, fs, 16)
        mism += int((st != sr).sum()); tot += B
        ok = (st == 1) & (sr == 1); okc += int(ok.sum())
        worst = max(worst, float(np.abs(tau[ok] - tr[ok]).max()))
        worstw = max(worstw, float(np.abs(wrn[ok] - wr[ok][:, :12]).max()))
    print(f"{name:9s} ({w.kernel_name()[:32]}) instances {tot}  status mismatches {mism}  ok {okc}  max|tau - oracle| {worst:.3e}  max|wrench - oracle| {worstw:.3e}", flush=True)
