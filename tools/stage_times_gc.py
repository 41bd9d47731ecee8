"""Diagnostic: stage breakdown (shader cycles) of the general-contact kernel from the DWBC_STAGE_TIMERS build
(`make -C libdwbc_amd/csrc timed`, run with DWBC_TIMED=1).   python tools/stage_times_gc.py [B] [flags e.g. 1110]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libdwbc_amd as D  # noqa: E402
from tests import cases  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
flags = [int(a) for a in (sys.argv[2] if len(sys.argv) > 2 else "1110")]
wbc = D.Batch(D.Model.from_urdf(cases.URDF), B)
for c in cases.CONTACTS_4:
    wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"])
wbc.add_task(0, D.TASK_LINK_6D, 0)
wbc.add_task(1, D.TASK_LINK_ROTATION, 15)
wbc.set_torque_limit(np.array(cases.TAU_LIM))
wbc.set_max_active_contacts(3)
q, _, fs = cases.synth_batch(B, seed=5, yaw=True)
wbc.set_state(q); wbc.set_contact(np.tile(np.array(flags, np.uint8), (B, 1))); wbc.set_fstar_all(fs)
for _ in range(3):
    wbc.solve()
wbc.sync()
print("kernel:", wbc.kernel_name(), "flags", flags)
nb = wbc._L.dwbc_batch_field_bytes(wbc._h, 13)  # the diag record is wider in the diagnostic build
d = np.zeros(nb // 4, dtype=np.int32)
wbc._L.dwbc_batch_get(wbc._h, 13, d.ctypes.data, nb)
d = d.reshape(B, -1)
DG_TIME = 14 + 5 * 12
names = ["kinematics + CRBA", "A^-1 (register sweep, 39)", "J_C, Y, Lambda_c (Gauss-Jordan 6 nc), Jbar", "A^-1 N_c", "Vb, NwJw, projector", "W^+ (register sweep, 33)",
         "FNl, gravity torque, P_C", "level 0: J_task, J_kt, chain", "level 0: QP inputs", "level 0: QP", "level 1: J_task, J_kt, chain", "level 1: QP inputs", "level 1: QP",
         "redistribution QP"]
t = np.median(d[:, DG_TIME:DG_TIME + 14], axis=0)
prev = 0
for i, nm in enumerate(names):
    if t[i] > 0:
        print(f"  {nm:46s} {t[i] - prev:9.0f}   (cumulative {t[i]:9.0f})")
        prev = t[i]
print("QP iterations (median) level 0 / 1 / redistribution:", np.median(d[:, 4]), np.median(d[:, 5]), np.median(d[:, 8]))
