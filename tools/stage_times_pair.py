"""Diagnostic: timeline of the paired (two-wave) kernel of dwbc_cycle2p.h from the DWBC_STAGE_TIMERS build (make -C libdwbc_amd/csrc
timed; DWBC_TIMED=1): when the main and the helper wave reach each of the five workgroup barriers, when the main wave leaves it
(the later of the two plus the barrier itself), and the main wave's stamps of the last phase.  Shader cycles since kernel start,
medians over the batch."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libdwbc_amd as D  # noqa: E402
from libdwbc_amd import workloads as W  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
wbc = D.Batch(D.Model.from_urdf(W.URDF), B)
for c in W.CONTACTS_2:
    wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"])
wbc.add_task(0, D.TASK_LINK_6D, 0)
wbc.add_task(1, D.TASK_LINK_ROTATION, 15)
wbc.set_torque_limit(np.array(W.TAU_LIM))
q, fl, fs = W.synth_batch(B, seed=20251226 + 2)
wbc.set_state(q); wbc.set_contact(fl); wbc.set_fstar_all(fs)
for _ in range(3):
    wbc.solve()
wbc.sync()
print("kernel:", wbc.kernel_name())
nb = wbc._L.dwbc_batch_field_bytes(wbc._h, 13)
d = np.zeros(nb // 4, dtype=np.int32)
wbc._L.dwbc_batch_get(wbc._h, 13, d.ctypes.data, nb)
d = d.reshape(B, -1)
t = np.median(d[:, 74:90].astype(np.float64), axis=0)      # DG_TIME: main arrivals 0..4, stamps 5.., 15
f = np.median(d[:, 90:90 + 64].astype(np.float64), axis=0)  # DG_FTIME: helper arrivals 0..4, main departures 8..12
names = ["B0 link frames", "B1 A^-1, riding rows | J_C, Vb, VG, Hb", "B2 Lambda_c, Jbar^T | D, Gram blocks", "B3 (gone)", "B4 W^+ a, T1 | Lambda_t, NwJw"]
print(f"{'barrier':32s} {'main arrives':>13s} {'helper arrives':>15s} {'main leaves':>12s} {'main waited':>12s}")
prev = 0.0
for i, n in enumerate(names):
    if t[i] == 0 and f[i] == 0:
        continue  # (a barrier this build does not have)
    print(f"{n:32s} {t[i]:13.0f} {f[i]:15.0f} {f[8 + i]:12.0f} {f[8 + i] - t[i]:12.0f}   (main worked {t[i] - prev:.0f} since the last barrier)")
    prev = f[8 + i]
for i, n in ((5, "stage 3a (J_kt, X, null-space chain)"), (6, "wrench maps (MFMA), QP set-up"), (7, "level-0 QP"), (8, "level-1 QP"), (15, "redistribution QP + outputs")):
    if t[i] > 0:
        print(f"{n:44s} {t[i] - prev:10.0f} {t[i]:12.0f}")
        prev = t[i]
fine = {51: "phase 1: world inertias", 52: "phase 1: composite inertias (both waves), S", 53: "phase 1: F", 54: "phase 1: mass-matrix pairs staged", 41: "phase 1: CRBA done (A^-1 sweep starts)", 42: "phase 2: Y = J_C A^-1 stored", 43: "phase 2: Lambda_c", 45: "phase 4: D Lambda_c of every slot, P_C", 46: "phase 4: cv (base residuals)", 47: "phase 4: E = Lambda_c d - lam", 48: "phase 4: tau_any", 49: "phase 4: (I - P) tau_any",
        16: "level-0 J_kt / X rows in registers", 17: "level-0 chain done", 18: "level-1 J_kt / X rows in registers", 19: "level-1 chain done", 24: "level-0 QP: rows ready", 25: "level-0 QP: committed",
        26: "redistribution QP: rows ready", 27: "redistribution QP: committed", 28: "torques stored"}
print("fine stamps of the main wave (cumulative):")
for i in sorted(fine, key=lambda i_: f[i_]):
    if f[i] > 0:
        print(f"  {fine[i]:44s} {f[i]:12.0f}")
qn = ["post-loop", "slack + arg-min", "publish n, r, z", "step / drop", "commit", "lexicographic point (CG)", "normalise rows, init", "feasibility of the point", "row fill"]
print("level-0 QP solver sections (cycles, summed over iterations):")
for i, n in enumerate(qn):
    print(f"  {n:28s} {f[32 + i]:10.0f}")
