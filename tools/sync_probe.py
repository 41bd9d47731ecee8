import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import torch
import bench
class A: pass
args = A(); args.batch=1024; args.dtype="f64"; args.workload="ds2"; args.no_hqp=False
eng = bench.HipEngine(args, 0, 0)
def run(mode, K=20):
    for _ in range(5): eng.solve()
    eng.synchronize(); eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): eng.solve()
    if mode == "spin":
        ev = torch.cuda.Event(); ev.record(eng.stream)
        while not ev.query(): pass
    eng.synchronize(); eng.synchronize()
    return (time.perf_counter() - t0) / K * 1e6
for mode in ("plain", "plain", "spin"):
    r = [run(mode) for _ in range(8)]
    print(mode, "us/step:", " ".join("%.1f" % x for x in r))
