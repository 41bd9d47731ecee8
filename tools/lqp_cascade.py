"""The TOCABI LQP cascade (ConfigureLQP + CalcControlTorqueLQP: 51 variables, 4 levels) at B = 1024 on the generic hierarchical-QP
solver -- the workload of tools/profile_lqp.sh (rocprofv3 kernel stats + counters of dwbc_hqp_kernel).  Prints the wall-clock rate,
the per-level iteration counts and a flop model of one cascade (so that the rocprof time can be put against the fp64 roof)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libdwbc_amd as D  # noqa: E402
from libdwbc_amd import hqp as Hq  # noqa: E402
from libdwbc_amd import workloads as W  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
wbc = D.Batch(D.Model.from_urdf(W.URDF), B, device=0)
for c in W.CONTACTS_2:
    wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
wbc.add_task(0, D.TASK_LINK_6D, 0)
wbc.add_task(1, D.TASK_LINK_ROTATION, 15)
wbc.set_torque_limit(np.array(W.TAU_LIM))
wbc.enable_dump(True)
q, fl, fs = W.synth_batch(B, seed=5)
wbc.set_state(q); wbc.set_contact(fl); wbc.set_fstar_all(fs)
wbc.solve(); wbc.sync()
hq = D.HQP.for_lqp(wbc, 12)
hq.configure_lqp(wbc); hq.solveSequential(); wbc.sync()
t0 = time.perf_counter()
for _ in range(reps):
    hq.configure_lqp(wbc)
    hq.solveSequential()
wbc.sync()
dt = (time.perf_counter() - t0) / reps
nv = 51
m = [66, 86, 0, 0]        # inequality rows per level (torque limits; cones + acceleration limits)
e = [6, 12, 6, 3]         # equality rows per level
ns = [45, 33, 27, 24]     # null-space sizes after each level (asserted in tests/test_hqp.py)
it = [float(np.mean(hq.get(lv, Hq.ITER))) for lv in range(4)]
# flop model of one cascade, per instance (dense formulas, 2 flop per multiply-add):
#   per level i with k = null-space size BEFORE the level (51, 45, 33, 27): Bz = B Z (e x nv x k), H = Bz^T Bz (+ Z^T Hc Z: nv x nv x k + nv x k x k),
#   H^-1 by Gauss-Jordan (2 k^3), C Z for the rows in play (rows x nv x k), per active-set step ~ 2 (rows x k) + 4 k q + q^3 / 3,
#   null-space extension by Householder QR of (B Z)^T and Z <- Z Q (2 nv k e)
kprev = [51, 45, 33, 27]
rows = [66, 66 + 86, 66 + 86, 66 + 86]
flop = 0.0
for i in range(4):
    k = kprev[i]
    flop += 2 * e[i] * nv * k + 2 * e[i] * k * k + (2 * nv * nv * k + 2 * nv * k * k if i >= 1 else 0) + 2 * k ** 3
    flop += 2 * rows[i] * nv * k
    steps = it[i] if it[i] == it[i] else 10.0
    flop += steps * (2 * rows[i] * k + 4 * k * 16 + 16 ** 3 / 3)
    flop += 2 * nv * k * e[i] + 2 * e[i] * e[i] * k
print(f"B = {B}: configure + cascade {dt * 1e3:.3f} ms per batch -> {B / dt:.0f} instances/s; active-set steps per level (mean) {it}")
print(f"flop model: {flop / 1e6:.2f} Mflop per instance -> {flop * B / dt / 1e12:.3f} TFLOP/s = {flop * B / dt / 78.6e12 * 100:.2f} % of the fp64 roof")
