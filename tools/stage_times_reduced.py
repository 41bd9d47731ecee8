"""Diagnostic: per-stage shader-cycle breakdown of the reduced-dynamics kernel (DWBC_STAGE_TIMERS build, DWBC_TIMED=1)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libdwbc_amd as D  # noqa: E402
from tests import cases  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = D.Model.from_urdf(cases.URDF)
wbc = D.Batch(model, B)
for c in cases.CONTACTS_2:
    wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"])
wbc.add_task(0, D.TASK_LINK_6D, 0)
wbc.add_task(1, D.TASK_LINK_ROTATION, 15)
q, fl, fs = cases.synth_batch(B, seed=20251226 + 2)
wbc.set_state(q); wbc.set_contact(fl); wbc.set_fstar_all(fs)
for _ in range(3):
    wbc.solve(reduced=True)
wbc.sync()
nb = wbc._L.dwbc_batch_field_bytes(wbc._h, 13)  # (the diagnostic build's record is longer than the product build's)
d = np.zeros(nb // 4, dtype=np.int32)
wbc._L.dwbc_batch_get(wbc._h, 13, d.ctypes.data, nb)
d = d.reshape(B, -1)
t = np.median(d[:, 74:90].astype(np.float64), axis=0)
order = [(0, "kin+CRBA"), (1, "A_inv"), (3, "reduced dynamics (J_I_nc, A_R, J_I_nc_inv_T)"), (2, "J_C/Lambda_c/Jbar/AiNc"),
         (4, "A_R_inv N_CR, J_CR_INV_T, G_R"), (5, "NwJw_R"), (6, "W_R^+ + grav"), (7, "J_base_R_kt"), (8, "task-space levels"),
         (9, "cascade + NC QP"), (10, "redistribution QP")]
prev = 0.0
print("stage                                          median cycles   cumulative")
for i, n in order:
    print(f"{n:46s} {t[i]-prev:12.0f} {t[i]:12.0f}")
    prev = t[i]
print("qp iters median (levels.., NC slot 3, redis slot 4)", np.median(d[:, 4:9], axis=0))
