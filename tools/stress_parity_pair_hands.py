"""Parity sweep of the two-wave kernel (B = 1024 per launch) on contact PAIRS that include the hands (four registered contacts, two
active per instance, random pair per instance, tilted base): the inverse-dynamics form of W^+ and the Jacobi-scaled Lambda_c on
foot + hand and hand + hand pairs, against oracle/dwbc_oracle.c.  Development / profiles only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import libdwbc_amd as D  # noqa: E402
from oracle import orc  # noqa: E402
from tests import cases  # noqa: E402

M = orc.make_model(cases.tocabi_model())
PAIRS = [[1, 1, 0, 0], [1, 0, 1, 0], [1, 0, 0, 1], [0, 1, 1, 0], [0, 1, 0, 1], [0, 0, 1, 1]]
NS = int(os.environ.get("STRESS_SEEDS", "4"))
B = 1024
w = D.Batch(D.Model.from_urdf(cases.URDF), B, device=0)
for c in cases.CONTACTS_4:
    w.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
for lv, links in enumerate(cases.TASKS_2LEVEL):
    for mode, link, pt in links:
        w.add_task(lv, mode, link, pt)
w.set_torque_limit(np.array(cases.TAU_LIM))
S = orc.make_setup(cases.CONTACTS_4, cases.TASKS_2LEVEL, cases.TAU_LIM)
worst = 0.0; mism = 0; tot = 0; okc = 0
for seed in range(NS):
    q, _, fs = cases.synth_batch(B, seed=9100 + seed, yaw=True)
    rng = np.random.default_rng(seed)
    fl = np.array([PAIRS[i] for i in rng.integers(0, len(PAIRS), B)], np.uint8)
    w.set_state(q); w.set_contact(fl); w.set_fstar_all(fs); w.solve()
    tau, st = w.get("tau"), w.get("status")
    tr, wr, sr, _ = orc.cycle_batch(M, S, q, fl, fs, 16)
    mism += int((st != sr).sum()); tot += B
    ok = (st == 1) & (sr == 1); okc += int(ok.sum())
    worst = max(worst, float(np.abs(tau[ok] - tr[ok]).max()))
print(f"hand_pairs ({w.kernel_name()[:40]}) instances {tot}  status mismatches {mism}  ok {okc}  max|tau - oracle| {worst:.3e}", flush=True)
