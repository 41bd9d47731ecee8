"""Randomised parity sweep of the kernel packs (models of other sizes) against the numpy restatement: every instance compared.
Development / profiles only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libdwbc_amd as D  # noqa: E402
from oracle import urdf_model  # noqa: E402
from tests import cases  # noqa: E402
from tests.test_model_packs import VARIANTS, model_43, oracle_cycle, states_43, variant_states, variant_urdf  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for name in ("fixed_arms", "fixed_head", "plus4"):
    if name == "plus4":
        md, mo = model_43()
        q, fs = states_43(B, 21)
        links = [6, 12, 15]
    else:
        path = variant_urdf(f"/tmp/{name}.urdf", VARIANTS[name][0])
        mo = urdf_model.load_urdf(path)
        md = D.Model.from_urdf(path)
        q, fs = variant_states(mo, B, seed=21)
        links = [md.link_id("L_AnkleRoll_Link"), md.link_id("R_AnkleRoll_Link"), md.link_id("Upperbody_Link")]
    q[:, 6:md.ndof] += 0.05 * np.random.default_rng(4).uniform(-1, 1, size=(B, md.ndof - 6))  # wider than the tests
    fs *= np.random.default_rng(5).uniform(0.5, 2.5, size=(B, 1))
    cases.ensure_pack(md)
    lim = np.full(md.ndof - 6, 300.0)
    wbc = D.Batch(md, B, device=0)
    for cc, l in zip(cases.CONTACTS_2, links[:2]):
        wbc.add_contact(l, cc["point"], cc["lx"], cc["ly"], cc["mu"], cc["muz"])
    wbc.add_task(0, D.TASK_LINK_6D, 0)
    wbc.add_task(1, D.TASK_LINK_ROTATION, links[2])
    wbc.set_torque_limit(lim)
    wbc.set_state(q); wbc.set_contact(np.ones((B, 2), np.uint8)); wbc.set_fstar_all(fs)
    wbc.solve()
    tau, st = wbc.get("tau"), wbc.get("status")
    mism, worst, ok = 0, 0.0, 0
    for b in range(B):
        o = oracle_cycle(mo, links, q[b], fs[b], lim)
        if st[b] != o["status"]:
            mism += 1
            continue
        if st[b]:
            ok += 1
            worst = max(worst, float(np.abs(tau[b] - np.stack([o["tau_grav"], o["tau_task"], o["tau_contact"]])).max()))
    print(f"{name:10s} ({md.ndof} dof)  instances {B}  status mismatches {mism}  ok {ok}  max|tau - oracle| {worst:.3e}")
