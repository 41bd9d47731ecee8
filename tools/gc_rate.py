import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import libdwbc_amd as D
from tests import cases
for B, flags in ((1024, [1, 1, 1, 0]), (1024, [1, 1, 0, 0]), (8192, [1, 1, 1, 0])):
    wbc = D.Batch(D.Model.from_urdf(cases.URDF), B, device=0)
    for c in cases.CONTACTS_4:
        wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    wbc.add_task(0, D.TASK_LINK_6D, 0)
    wbc.add_task(1, D.TASK_LINK_ROTATION, 15)
    wbc.set_torque_limit(np.array(cases.TAU_LIM))
    wbc.set_max_active_contacts(3)
    q, _, fs = cases.synth_batch(B, seed=5, yaw=True)
    wbc.set_state(q); wbc.set_contact(np.tile(np.array(flags, np.uint8), (B, 1))); wbc.set_fstar_all(fs)
    wbc.solve(); wbc.sync()
    ms = wbc.time_solves(20) / 20
    print(f"general-contact kernel, B = {B}, flags {flags} ({wbc.kernel_name()}): {ms:8.3f} ms per launch -> {B / ms * 1e3 / 1e6:6.3f} M cycles/s, status ok {wbc.get('status').mean():.3f}", flush=True)
