"""CPU-side parity sweep of the two-wave kernel SOURCE (dwbc_cycle2p.h in the host emulation, both roles run in turn) against
oracle/dwbc_oracle.c on the two-level configurations it is built for: torques, wrench and status.  Development aid, no GPU needed."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc  # noqa: E402
from tests import cases  # noqa: E402
from tests.emu.emu import Emu  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
M = orc.make_model(cases.tocabi_model())
cfgs = {"ds": (cases.TASKS_2LEVEL, {}, cases.TAU_LIM), "ds_yaw": (cases.TASKS_2LEVEL, dict(yaw=True), cases.TAU_LIM),
        "mixed": (cases.TASKS_2LEVEL, dict(contact_mode="mixed"), cases.TAU_LIM), "ds_nolim": (cases.TASKS_2LEVEL, {}, None)}
for name, (tasks, kw, lim) in cfgs.items():
    e = Emu(cases.URDF, cases.CONTACTS_2, tasks, lim)
    S = orc.make_setup(cases.CONTACTS_2, tasks, lim)
    q, fl, fs = cases.synth_batch(B, seed=9000, **kw)
    r = e.run(q, fl, fs, compact="pair")
    tr, wr, sr, _ = orc.cycle_batch(M, S, q, fl, fs, 8)
    st = r["status"]
    ok = (st == 1) & (sr == 1)
    err = np.abs(r["tau"][ok] - tr[ok]).max(axis=(1, 2))
    werr = np.abs(r["wrench"][ok] - wr[ok][:, :12]).max()
    print(f"{name:9s} B {B} status mismatches {int((st != sr).sum())} ok {int(ok.sum())} max|tau - oracle| {err.max():.3e} p99 {np.percentile(err, 99):.2e} "
          f"wrench {werr:.2e} iters {r['diag'][:, 4:9].sum(axis=0)}", flush=True)
