"""CPU-side parity sweep of the general-contact kernel source (tests/emu, dwbc_cycle_gc.h) against oracle/dwbc_oracle.c: random contact sets (one to three
active contacts, hands included), two and three task levels.  python tools/stress_emu_gc.py"""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases
from tests.emu.emu import Emu
from oracle import orc
M = orc.make_model(cases.tocabi_model())
B = 1024
sets = [[1, 1, 1, 0], [1, 1, 0, 1], [1, 0, 1, 1], [0, 1, 1, 1], [1, 1, 0, 0], [1, 0, 0, 0], [0, 1, 0, 0], [1, 0, 1, 0], [0, 1, 0, 1], [0, 0, 1, 1]]
worst = 0
for tasks, kw, nm in ((cases.TASKS_2LEVEL, dict(yaw=True), "2lv"), (cases.TASKS_3LEVEL_SWING_R, dict(yaw=True, contact_mode="L", levels=3), "3lv")):
    S = orc.make_setup(cases.CONTACTS_4, tasks, cases.TAU_LIM)
    e = Emu(cases.URDF, cases.CONTACTS_4, tasks, cases.TAU_LIM)
    for seed in range(4):
        q, _, fs = cases.synth_batch(B, seed=7000 + seed, **kw)
        rng = np.random.default_rng(100 + seed)
        ss = sets if nm == "2lv" else [[1, 0, 1, 1], [1, 0, 1, 0], [1, 0, 0, 1], [1, 0, 0, 0]]
        fl = np.array([ss[i] for i in rng.integers(0, len(ss), B)], np.uint8)
        tr, wr, sr, _ = orc.cycle_batch(M, S, q, fl, fs, 4)
        r = e.run_gc(q, fl, fs)
        ok = (sr == 1) & (r["status"] == 1)
        d = np.abs(r["tau"] - tr).max(axis=(1, 2))
        bad = np.where(ok & (d > 1e-6))[0]
        print(nm, "seed", seed, "status mismatches", int((sr != r["status"]).sum()), "ok", int(ok.sum()), "max %.3e" % d[ok].max(), "bad", [(int(i), fl[i].tolist(), float(d[i])) for i in bad[:5]], flush=True)
