"""PCIe-inclusive rate of the host-array boundary (dwbc_batch_set_state / set_contact / set_fstar -> solve -> get):
what a caller pays when its states live in host memory.  Never reported as bench.py's `value`.
Two callers: one that hands over its own (pageable) arrays, one that assembles its states in the batch's page-locked mirrors
(Batch.host_view / dwbc_batch_host_ptr) so that the host-side copy disappears."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libdwbc_amd as D  # noqa: E402
from tests import cases  # noqa: E402

for B in (1024, 8192, 65536):
    q, fl, fs = cases.synth_batch(B, seed=3)
    wbc = D.Batch(D.Model.from_urdf(cases.URDF), B, device=0)
    for c in cases.CONTACTS_2:
        wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    wbc.add_task(0, D.TASK_LINK_6D, 0)
    wbc.add_task(1, D.TASK_LINK_ROTATION, 15)
    wbc.set_torque_limit(np.array(cases.TAU_LIM))
    for _ in range(3):
        wbc.set_state(q); wbc.set_contact(fl); wbc.set_fstar_all(fs); wbc.solve(); wbc.get("tau_total"); wbc.get("status")
    K = 50 if B <= 8192 else 10
    t0 = time.perf_counter()
    for _ in range(K):
        wbc.set_state(q); wbc.set_contact(fl); wbc.set_fstar_all(fs); wbc.solve(); tau = wbc.get("tau_total"); st = wbc.get("status")
    dt = (time.perf_counter() - t0) / K
    # the same with the states written into the page-locked mirrors (what a simulator loop would do)
    vq, vf, vs = wbc.host_view("in_q"), wbc.host_view("in_contact"), wbc.host_view("in_fstar")
    vq[:], vf[:], vs[:] = q, fl, fs
    t0 = time.perf_counter()
    for _ in range(K):
        vq[0, 0] += 0.0  # (the caller's own writes would go here)
        wbc.set_state(vq); wbc.set_contact(vf); wbc.set_fstar_all(vs); wbc.solve(); tau2 = wbc.get("tau_total"); st2 = wbc.get("status")
    dt2 = (time.perf_counter() - t0) / K
    assert np.array_equal(tau, tau2) and np.array_equal(st, st2)
    print(f"B={B}: {dt*1e3:.3f} ms per host-to-host cycle batch  ->  {B/dt/1e6:.2f} M cycles/s PCIe-inclusive "
          f"(in {q.nbytes + fl.nbytes + fs.nbytes} B, out {tau.nbytes + st.nbytes} B); states assembled in the page-locked mirrors: "
          f"{dt2*1e3:.3f} ms -> {B/dt2/1e6:.2f} M cycles/s")
