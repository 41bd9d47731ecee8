"""Diagnostic: per-stage shader-cycle breakdown from the DWBC_STAGE_TIMERS build (run with DWBC_TIMED=1)."""
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
wbc.set_torque_limit(np.array(cases.TAU_LIM))
q, fl, fs = cases.synth_batch(B, seed=20251226 + 2)
wbc.enable_dump(True)
wbc.set_state(q); wbc.set_contact(fl); wbc.set_fstar_all(fs)
for _ in range(3):
    wbc.solve()
wbc.sync()
d = wbc.get("diag")
t = d[:, 74:90].astype(np.float64)
names = ["kin+CRBA", "A_inv", "JC/Lam/Jbar/AiNc", "NwJw+proj", "W_inv", "grav", "L0 jkt", "L0 qp rows", "L0 qp solve",
         "L1 jkt", "L1 qp rows", "L1 qp solve", "L2 jkt", "L2 rows", "(to redis)", "redis qp"]
med = np.median(t, axis=0)
prev = 0.0
print("stage                      median cycles   cumulative")
for i, n in enumerate(names):
    if med[i] <= 0:
        continue
    print(f"{n:26s} {med[i]-prev:12.0f} {med[i]:12.0f}")
    prev = med[i]
print("diag stamps 12,13 (cumulative):", med[12], med[13])
print("qp iters median", np.median(d[:, 4:9], axis=0), "nact", np.median(d[:, 9:14], axis=0))

import ctypes
nb = wbc._L.dwbc_batch_field_bytes(wbc._h, 49)
raw = np.zeros(nb // 8)
wbc._L.dwbc_batch_get(wbc._h, 49, raw.ctypes.data, nb)
raw = raw.reshape(B, -1)
st = np.median(raw[:, -64:], axis=0)
fn = ["start stage1", "J_C", "Y", "Lambda_c", "JbT", "AiNc", "vec,PC", "Vb", "JV", "gj6", "NwJw", "gram+inv", "VG", "FNl", "L0 Jt+T1", "L0 JAJ", "L0 Lambda_t", "all levels", "W+aP", "W sweep", "W corr+grav", "L0 Q,QW", "L0 QWQ inv"]
prev = st[0]
print("fine stamps (dump enabled, so absolute values include dump stores):")
for i, n in enumerate(fn):
    print(f"  {n:16s} {st[i]-prev:10.0f} {st[i]:10.0f}")
    prev = st[i]
qn = ["post-loop", "slack+argmin", "publish n, r, z", "step/drop", "commit", "rows+QR", "R^T y", "reflect+feas"]
print("level-0 QP solver sections (cycles, summed over iterations):")
for i, n in enumerate(qn):
    print(f"  {n:16s} {st[23 + i]:10.0f}")
kn = ["q load", "local rotations", "FK levels", "world inertias", "composite inertias", "S axes", "F = Ic S", "zero A", "CRBA walk", "A -> registers"]
print("kinematics sections (cycles):")
prev = 0.0
for i, n in enumerate(kn):
    print(f"  {n:20s} {st[32 + i]-prev:10.0f} {st[32 + i]:10.0f}")
    prev = st[32 + i]
