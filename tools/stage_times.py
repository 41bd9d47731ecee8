"""Diagnostic: per-stage shader-cycle breakdown of the fused cycle kernel from the DWBC_STAGE_TIMERS build
(`make -C libdwbc_amd/csrc timed`, run with DWBC_TIMED=1).

Default: the LEAN kernel (the one bench.py times; dump off): coarse stamps, then the fine-grained ones, the QP solver sections
and the kinematics sections (all of them live in the diag record of the diagnostic build).  `--full` switches the dump record
on (the full build, EXTRAS = true): those numbers include the dump stores and the optional paths' register pressure.

Stamps are cumulative cycle counts taken at fixed points of the kernel; the task-space stamps of ALL levels are taken before the
QP cascade starts, so the table is printed in CHRONOLOGICAL order (sorted by the median stamp) and stamps that a configuration
never writes (value 0) are left out -- differences between neighbours are then always non-negative."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libdwbc_amd as D  # noqa: E402
from tests import cases  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
full = "--full" in sys.argv
B = int(args[0]) if args else 1024
model = D.Model.from_urdf(cases.URDF)
wbc = D.Batch(model, B)
for c in cases.CONTACTS_2:
    wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"])
wbc.add_task(0, D.TASK_LINK_6D, 0)
wbc.add_task(1, D.TASK_LINK_ROTATION, 15)
wbc.set_torque_limit(np.array(cases.TAU_LIM))
q, fl, fs = cases.synth_batch(B, seed=20251226 + 2)
if full:
    wbc.enable_dump(True)
wbc.set_state(q); wbc.set_contact(fl); wbc.set_fstar_all(fs)
for _ in range(3):
    wbc.solve()
wbc.sync()
print("kernel:", wbc.kernel_name())
nb = wbc._L.dwbc_batch_field_bytes(wbc._h, 13)
d = np.zeros(nb // 4, dtype=np.int32)
wbc._L.dwbc_batch_get(wbc._h, 13, d.ctypes.data, nb)
d = d.reshape(B, -1)
t = d[:, 74:90].astype(np.float64)
# stamp index -> what has finished when it is taken (dwbc_cycle2.h / stage0 / stage1: DWBC_STAMP(i))
names = {0: "kinematics + CRBA", 1: "A^-1 sweep", 2: "J_C, Lambda_c, Jbar, A^-1 N_c", 12: "NwJw, projector on null(W)", 13: "level-0 J_t, T1",
         3: "task Jacobians, Lambda_t of all levels", 4: "W^+ sweep + gravity torque", 6: "level-0 J_kt / null space", 9: "level-1 J_kt / null space",
         7: "level-0 QP rows", 8: "level-0 QP solve", 10: "level-1 QP rows", 11: "level-1 QP solve", 15: "redistribution QP + outputs"}
med = np.median(t, axis=0)
rows = sorted((med[i], names[i]) for i in names if med[i] > 0)
prev = 0.0
print(f"{'stage (chronological)':44s} {'cycles':>10s} {'cumulative':>12s}")
for v, n in rows:
    print(f"{n:44s} {v - prev:10.0f} {v:12.0f}")
    prev = v
print("QP iterations (median) level 0 / level 1 / redistribution:", np.median(d[:, 4], axis=0), np.median(d[:, 5], axis=0), np.median(d[:, 8], axis=0),
      "| working-set sizes:", np.median(d[:, 9], axis=0), np.median(d[:, 10], axis=0), np.median(d[:, 13], axis=0))

if d.shape[1] >= 90 + 64:
    st = np.median(d[:, 90:90 + 64].astype(np.float64), axis=0)
    fn = {0: "stage 1 starts", 1: "J_C", 2: "Y = J_C A^-1", 3: "Lambda_c", 5: "Jbar^T, A^-1 N_c update", 6: "gravity pre-vector, P_C", 7: "Vb", 8: "Jbar Vb",
          12: "Gram matrix, NwJw, VG", 13: "FNl", 14: "level-0 J_t + T1", 15: "level-0 J A J^T", 16: "level-0 Lambda_t", 17: "all task levels",
          18: "W + alpha P", 19: "W sweep", 20: "W^+ correction + gravity torque", 21: "level-0 Q, Q W^+", 22: "level-0 Q W^+ Q^T inverse",
          42: "level-0 QP: base torque and wrench", 43: "wrench maps of the cascade (MFMA)"}
    rows = sorted((st[i], fn[i]) for i in fn if st[i] > 0)
    prev = rows[0][0] if rows else 0.0
    print("fine stamps:")
    for v, n in rows:
        print(f"  {n:36s} {v - prev:10.0f} {v:12.0f}")
        prev = v
    qn = ["post-loop", "slack + arg-min", "publish n, r, z", "step / drop", "commit", "lexicographic point (CG)", "normalise rows, init", "feasibility of the point"]
    print("level-0 QP solver sections (cycles, summed over iterations):")
    for i, n in enumerate(qn):
        print(f"  {n:24s} {st[23 + i]:10.0f}")
    print(f"  {'row fill (before the solver)':24s} {st[44]:10.0f}")
    kn = ["q load", "local rotations", "FK rounds", "world inertias", "composite inertias", "S axes", "F = Ic S", "zero A", "CRBA pairs", "A -> registers"]
    print("kinematics sections (cycles):")
    prev = 0.0
    for i, n in enumerate(kn):
        if st[32 + i] > 0:
            print(f"  {n:24s} {st[32 + i] - prev:10.0f} {st[32 + i]:12.0f}")
            prev = st[32 + i]
