"""Development probe: instances of the three-contact parity sweep (tools/stress_parity.py, gc_any) that differ from the oracle by more than 1e-6."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import libdwbc_amd as D
from tests import cases
from oracle import orc
M = orc.make_model(cases.tocabi_model())
S = orc.make_setup(cases.CONTACTS_4, cases.TASKS_2LEVEL, cases.TAU_LIM)
B = 2048
sets = [[1, 1, 1, 0], [1, 1, 0, 1], [1, 0, 1, 1], [0, 1, 1, 1], [1, 1, 0, 0], [1, 0, 0, 0], [0, 1, 0, 0], [1, 0, 1, 0], [0, 1, 0, 1]]
w = D.Batch(D.Model.from_urdf(cases.URDF), B, device=0)
for c in cases.CONTACTS_4: w.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
for lv, links in enumerate(cases.TASKS_2LEVEL):
    for mode, link, pt in links: w.add_task(lv, mode, link, pt)
w.set_torque_limit(np.array(cases.TAU_LIM)); w.set_max_active_contacts(3)
import sys as _s
NS = int(_s.argv[1]) if len(_s.argv) > 1 else 3
if len(_s.argv) > 2: sets = [[int(a) for a in _s.argv[2]]]
for seed in range(NS):
    q, _, fs = cases.synth_batch(B, seed=9100 + seed, yaw=True)
    rng = np.random.default_rng(seed)
    fl = np.array([sets[i] for i in rng.integers(0, len(sets), B)], np.uint8)
    w.set_state(q); w.set_contact(fl); w.set_fstar_all(fs); w.solve()
    tau, st, dg = w.get("tau"), w.get("status"), w.get("diag")
    tr, wr, sr, _ = orc.cycle_batch(M, S, q, fl, fs, 16)
    d = np.abs(tau - tr).max(axis=2)
    bad = np.where(d.max(axis=1) > 1e-6)[0]
    print("seed", seed, "bad", len(bad))
    for i in bad[:10]:
        print("  inst", i, "flags", fl[i], "d grav/task/contact %.2e %.2e %.2e" % tuple(d[i]), "iters", dg[i, 4:6], "nact", dg[i, 9:11], "redis", dg[i, 8], dg[i, 13], "st", st[i], sr[i])
