"""Development probe: one instance of the three-contact sweep through a tracing build (make experiment VARIANT=tr XFLAGS=-DDWBC_QP_TRACE)."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import libdwbc_amd as D
from tests import cases
seed, i = int(sys.argv[1]), int(sys.argv[2])
flags = [int(a) for a in sys.argv[3]]
q, _, fs = cases.synth_batch(2048, seed=9100 + seed, yaw=True)
w = D.Batch(D.Model.from_urdf(cases.URDF), 1, device=0)
for c in cases.CONTACTS_4: w.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
for lv, links in enumerate(cases.TASKS_2LEVEL):
    for mode, link, pt in links: w.add_task(lv, mode, link, pt)
w.set_torque_limit(np.array(cases.TAU_LIM)); w.set_max_active_contacts(3)
w.set_state(q[i:i + 1]); w.set_contact(np.array([flags], np.uint8)); w.set_fstar_all(fs[i:i + 1]); w.solve(); w.sync()
print("status", w.get("status"))
