"""Randomised sweep of the batched HQP paths on the GPU (LQP cascade and JACC QPs): status rates and the constraints each
formulation must keep, over the synthetic state distributions of SURVEY 8d (nominal stance jitter, random yaw + tilt, larger
task accelerations).  Development / profiles only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libdwbc_amd as D  # noqa: E402
from libdwbc_amd import hqp as Hq  # noqa: E402
from tests import cases  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for name, kw, scale in (("ds", dict(seed=101), 1.0), ("ds_yaw", dict(seed=102, yaw=True), 1.0), ("ds_x3", dict(seed=103), 3.0), ("ds_yaw_x3", dict(seed=104, yaw=True), 3.0)):
    q, fl, fs = cases.synth_batch(B, **kw)
    fs = fs * scale
    wbc = D.Batch(D.Model.from_urdf(cases.URDF), B, device=0)
    for c in cases.CONTACTS_2:
        wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    wbc.add_task(0, D.TASK_LINK_6D, 0)
    wbc.add_task(1, D.TASK_LINK_ROTATION, 15)
    wbc.set_torque_limit(np.array(cases.TAU_LIM))
    wbc.enable_dump(True)
    wbc.set_state(q); wbc.set_contact(fl); wbc.set_fstar_all(fs)
    wbc.solve()
    A, JC, G = wbc.get("A"), wbc.get("J_C"), wbc.get("G")
    hq = D.HQP.for_lqp(wbc, 12)
    hq.configure_lqp(wbc)
    hq.solveSequential()
    st = np.stack([hq.get(lv, Hq.STATUS) for lv in range(4)], axis=1)
    it = np.stack([hq.get(lv, Hq.ITER) for lv in range(4)], axis=1)
    y = hq.y_ans(3)
    tau = hq.lqp_torque(wbc)
    ok = st.all(axis=1)
    B0, b0 = hq.get(0, Hq.MAT_B).reshape(B, 6, 51), hq.get(0, Hq.VEC_b)
    A1, a1, v1 = hq.get(1, Hq.MAT_A).reshape(B, 86, 51), hq.get(1, Hq.VEC_a), hq.v_ans(1)
    dyn = np.abs(np.einsum("bij,bj->bi", B0, y) + b0).max(axis=1)
    viol = (np.einsum("bij,bj->bi", A1, y) + a1 - v1).max(axis=1)
    print(f"{name:10s} LQP : ok {ok.mean():.4f}  iters max {it.max(axis=0)}  |base dynamics| max {dyn[ok].max():.2e}  level-1 row violation max {viol[ok].max():.2e}  "
          f"|tau| max {np.abs(tau[ok]).max():.1f}  slack used in {float((np.abs(v1[ok]).max(axis=1) > 1e-9).mean()):.3f} of the instances")
    res = []
    for lv in range(2):
        hq.solve_jacc(wbc, lv)
        res.append(D.HQP.jacc_result(wbc, lv))
    for lv in range(2):
        r = res[lv]
        okj = r["status"] == 1
        acc, tq, f = r["acc_qp"], r["torque_qp"], r["contact_qp"]
        dynj = np.einsum("bij,bj->bi", A, acc) + np.einsum("bji,bj->bi", JC, f) + G
        dynj[:, 6:] -= tq
        print(f"{name:10s} JACC level {lv}: ok {okj.mean():.4f}  |dynamics| max {np.abs(dynj[okj]).max():.2e}  |J_C qddot| max {np.abs(np.einsum('bij,bj->bi', JC, acc)[okj]).max():.2e}  "
              f"|qddot_joint| max {np.abs(acc[okj][:, 6:]).max():.4f}  |tau| max {np.abs(tq[okj]).max():.2f}")
