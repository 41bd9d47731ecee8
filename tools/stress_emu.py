"""CPU-side parity sweep of the kernel SOURCE (host emulation, tests/emu) against oracle/dwbc_oracle.c: the same five contact /
task configurations as tools/stress_parity.py, at a size a CPU finishes in a minute.  Development aid for changes to the QP
solver (no GPU needed); the GPU sweep stays the one that is recorded in profiles/."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc  # noqa: E402
from tests import cases  # noqa: E402
from tests.emu.emu import Emu  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
M = orc.make_model(cases.tocabi_model())
cfgs = {"ds": (cases.TASKS_2LEVEL, {}), "ds_yaw": (cases.TASKS_2LEVEL, dict(yaw=True)), "mixed": (cases.TASKS_2LEVEL, dict(contact_mode="mixed")),
        "ss_L": (cases.TASKS_3LEVEL_SWING_R, dict(contact_mode="L", levels=3)), "ss_R": (cases.TASKS_3LEVEL_SWING_L, dict(contact_mode="R", levels=3))}
for name, (tasks, kw) in cfgs.items():
    e = Emu(cases.URDF, cases.CONTACTS_2, tasks, cases.TAU_LIM)
    S = orc.make_setup(cases.CONTACTS_2, tasks, cases.TAU_LIM)
    q, fl, fs = cases.synth_batch(B, seed=9000, **kw)
    t0 = time.time()
    r = e.run(q, fl, fs)
    t1 = time.time()
    tr, wr, sr, _ = orc.cycle_batch(M, S, q, fl, fs, 8)
    st = r["status"]
    ok = (st == 1) & (sr == 1)
    err = np.abs(r["tau"][ok] - tr[ok]).max(axis=(1, 2))
    print(f"{name:7s} B {B} status mismatches {int((st != sr).sum())} ok {int(ok.sum())} max|tau - oracle| {err.max():.3e} p99 {np.percentile(err, 99):.2e} "
          f"iters {r['diag'][:, 4:9].sum(axis=0)} emu {t1 - t0:.1f}s", flush=True)
