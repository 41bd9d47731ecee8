"""Development probe (not a test): launch cost of the fused cycle in a Python loop vs a captured HIP graph.  Run on the GPU box."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import libdwbc_amd as D
from tests import cases


def main():
    B=1024
    model = D.Model.from_urdf(cases.URDF)
    wbc = D.Batch(model, B, device=0)
    for c in cases.CONTACTS_2: wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    wbc.add_task(0, D.TASK_LINK_6D, 0); wbc.add_task(1, D.TASK_LINK_ROTATION, 15)
    wbc.set_torque_limit(np.array(cases.TAU_LIM))
    q, flags, fstar = cases.synth_batch(B, seed=1)
    dev=torch.device("cuda:0")
    tq=torch.from_numpy(q).to(dev); tf=torch.from_numpy(flags).to(dev); ts=torch.from_numpy(fstar).to(dev)
    ttau=torch.zeros((B,3,33),dtype=torch.float64,device=dev); twr=torch.zeros((B,12),dtype=torch.float64,device=dev); tst=torch.zeros((B,),dtype=torch.int32,device=dev)
    for n,t in (("in_q",tq),("in_contact",tf),("in_fstar",ts),("tau",ttau),("wrench",twr),("status",tst)): wbc.bind_tensor(n,t)
    s=torch.cuda.current_stream(); wbc.set_stream(s.cuda_stream)
    for _ in range(5): wbc.solve()
    torch.cuda.synchronize()
    K=50
    t0=time.perf_counter()
    for _ in range(K): wbc.solve()
    torch.cuda.synchronize()
    print("loop   us/step", (time.perf_counter()-t0)/K*1e6)
    g=torch.cuda.CUDAGraph()
    cs=torch.cuda.Stream()
    with torch.cuda.stream(cs):
        wbc.set_stream(cs.cuda_stream)
        g.capture_begin()
        for _ in range(K): wbc.solve()
        g.capture_end()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    t0=time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    print("graph  us/step", (time.perf_counter()-t0)/K*1e6)
    ref=ttau.clone()
    print("status ok", float(tst.float().mean()))


if __name__ == "__main__":
    main()
