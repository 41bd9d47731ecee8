#!/usr/bin/env python3
"""bench.py -- HQP control cycles/s on batched TOCABI states (BASELINE.json metric), one process per GPU.

A "step" = one pass of the fused hot path (UpdateKinematics .. CalcContactRedistribute, one kernel launch) over one
batch of B synthetic TOCABI states that are already resident in HBM.  N = 1 workload = BASELINE.json configs[1]:
batch = 1024, double support, 2-level HQP (pelvis 6D, upper-body rotation), tau limit 300, fp64.
N > 1: every rank solves its own B instances (weak scaling, no data-path collective); the only collective is the
final RCCL all_gather of (tau[33], wrench[12], status) of the last step, inside the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F_ALG = 1.34e6          # flop per cycle, double support 2-level with tau limit (SURVEY 8d / BASELINE.md 3)
PEAK_FP64_TFLOPS = 78.6  # MI355X fp64 vector = matrix peak (public spec; MI355X_MICROARCH.md lists no fp64 row)
PEAK_FP32_TFLOPS = 157.3  # MI355X fp32 vector peak (public spec), for --dtype f32 runs
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r01_final_pmc_summary.json")  # rocprofv3 --pmc passes of this command


def host_cores():
    """CPU threads this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box hands one GPU's
    share of the host, 16 cores, to a job although all 256 logical CPUs are visible) -- oversubscribing the quota makes the
    OpenMP baseline slower, not faster."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return min(n, int(os.environ.get("DWBC_CPU_THREADS", "16")))


def hbm_traffic_per_launch(kernel_name, batch):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE, KiB -> bytes) if they were
    taken on the same kernel and batch; None otherwise.  The accesses are 8 B per lane, for which the gfx950 FETCH_SIZE
    correction of MI355X_MICROARCH.md (x2 for 16-B-per-lane streams) is not calibrated: the raw sum is reported."""
    try:
        pm = json.load(open(PMC_SUMMARY))
        k = pm["_kernel"]
        if not kernel_name.startswith(k["kernel"].split("<")[0].replace("void ", "")) or int(k["grid"]) != batch * 64:
            return None
        return (pm["FETCH_SIZE"]["mean_per_launch"] + pm["WRITE_SIZE"]["mean_per_launch"]) * 1024.0
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1024, help="instances per GPU")
    ap.add_argument("--workload", default="ds2", choices=["ds2", "ss3", "mixed", "reduced"],
                    help="ds2 = BASELINE configs[1] (the metric's config, default); ss3 / mixed / reduced = configs[2] / [3] / [4] "
                         "(parity-test cases; measured for DESIGN.md only)")
    ap.add_argument("--no-hqp", action="store_true", help="hqp = false: plain hierarchy + closed-form redistribution (DESIGN.md only)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"], help="arithmetic type of the kernels (f32: measured for DESIGN.md only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import torch

    import libdwbc_amd as D
    from tests import cases

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # DWBC_BENCH_BACKEND=gloo: rehearsal of the multi-rank code path on a box with fewer GPUs than ranks (ranks share
        # devices, collectives go through host tensors).  The driver's runs use the default, RCCL with one GPU per rank.
        backend = os.environ.get("DWBC_BENCH_BACKEND", "nccl")
        if backend != "nccl":
            local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    else:
        backend = None
        torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    cdev = dev if backend in (None, "nccl") else torch.device("cpu")  # where collective buffers live
    B = args.batch

    model = D.Model.from_urdf(cases.URDF)
    wbc = D.Batch(model, B, device=local_rank, dtype=args.dtype)
    for c in cases.CONTACTS_2:
        wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    wbc.add_task(0, D.TASK_LINK_6D, 0)
    wbc.add_task(1, D.TASK_LINK_ROTATION, 15)
    wl = args.workload
    reduced = wl == "reduced"
    if wl == "ss3":
        wbc.add_task(2, D.TASK_LINK_6D, 12)  # swing (right) foot, SURVEY 8d config 3
    if not reduced:
        wbc.set_torque_limit(np.array(cases.TAU_LIM))  # the reference's reduced path runs without the limit (App. C-10)
    f_alg = {"ds2": F_ALG, "ss3": 1.37e6, "mixed": F_ALG, "reduced": F_ALG}[wl]
    workload_name = {
        "ds2": "BASELINE configs[1]: batch=1024 per GPU, TOCABI double support, 2-level HQP (pelvis 6D + upper-body rotation), tau limit 300, fp64",
        "ss3": "BASELINE configs[2]: TOCABI left single support + swing-foot task (3-level HQP), tau limit 300, fp64",
        "mixed": "BASELINE configs[3]: TOCABI mixed contact modes LR/L/R = 1/2,1/4,1/4 per instance, 2-level HQP, tau limit 300, fp64",
        "reduced": "BASELINE configs[4] in fp64: TOCABI double support through the reduced (centroidal) dynamics path, 2-level HQP, no tau limit",
    }[wl]
    kw = {"ss3": dict(contact_mode="L", levels=3), "mixed": dict(contact_mode="mixed")}.get(wl, {})
    q, flags, fstar = cases.synth_batch(B, seed=20251226 + 2 + 1000 * rank, **kw)
    tq = torch.from_numpy(q).to(dev)
    tf = torch.from_numpy(flags).to(dev)
    ts = torch.from_numpy(fstar).to(dev)
    # outputs packed per instance as [tau(3x33) | wrench(12) | status] for the final gather
    ttau = torch.zeros((B, 3, 33), dtype=torch.float64, device=dev)
    twr = torch.zeros((B, 12), dtype=torch.float64, device=dev)
    tst = torch.zeros((B,), dtype=torch.int32, device=dev)
    for name, t in (("in_q", tq), ("in_contact", tf), ("in_fstar", ts), ("tau", ttau), ("wrench", twr), ("status", tst)):
        wbc.bind_tensor(name, t)
    stream = torch.cuda.current_stream()
    wbc.set_stream(stream.cuda_stream)

    def gather_final():
        if world == 1:
            return None
        pack = torch.cat([ttau.sum(dim=1), twr, tst.to(torch.float64)[:, None]], dim=1).contiguous()  # B x 46
        pack = pack.to(cdev)
        out = torch.empty((world * B, pack.shape[1]), dtype=torch.float64, device=cdev)
        dist.all_gather_into_tensor(out, pack)
        return out

    for _ in range(args.warmup):
        wbc.solve(hqp=not args.no_hqp, reduced=reduced)
    gather_final()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wbc.solve(hqp=not args.no_hqp, reduced=reduced)
    gather_final()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    status_ok = float(tst.float().mean().item())

    # kernel-only time with HIP events on the launch stream (roofline.achieved)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(stream)
    for _ in range(args.steps):
        wbc.solve(hqp=not args.no_hqp, reduced=reduced)
    ev1.record(stream)
    torch.cuda.synchronize()
    kern_ms = ev0.elapsed_time(ev1) / args.steps

    if rank == 0:
        value = world * B * args.steps / dt
        achieved = f_alg * B / (kern_ms * 1e-3) / 1e12
        nt, lds = wbc.launch_info()
        line = {
            "metric": "HQP control cycles/sec (batched TOCABI)",
            "value": value,
            "unit": "cycles/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": workload_name + (" [hqp = false]" if args.no_hqp else ""),
                "batch_per_gpu": B,
                "threads_per_instance": nt,
                "lds_bytes_per_instance": lds,
                "status_ok_fraction": status_ok,
            },
            "roofline": {
                "bound": "mfma",
                "achieved": achieved,
                "peak": PEAK_FP32_TFLOPS if args.dtype == "f32" else PEAK_FP64_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / (PEAK_FP32_TFLOPS if args.dtype == "f32" else PEAK_FP64_TFLOPS),
                "traffic": hbm_traffic_per_launch(wbc.kernel_name(), B),
                "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE + WRITE_SIZE, profiles/r01_final_pmc_summary.json)",
                "kernel": wbc.kernel_name(),
                "kernel_ms": kern_ms,
                "flop_per_cycle": f_alg,
                "note": "fp64 FMA roof (vector = matrix rate on MI355X, public spec 78.6 TFLOP/s); algorithmic flop of the "
                        "reference's dense formulas.  batch 1024 = one wave per SIMD, where tools/ubench measures 23.3 TFLOP/s "
                        "for back-to-back fp64 FMAs from a single wave (24.0 for MFMA f64 16x16x4): DESIGN.md 'Measured'",
            },
        }
        if world == 1 and not args.no_cpu_baseline and wl == "ds2":
            from oracle import orc

            M = orc.make_model(cases.tocabi_model())
            S = orc.make_setup(cases.CONTACTS_2, cases.TASKS_2LEVEL, cases.TAU_LIM)
            cores = host_cores()
            qs, fls, fss = cases.synth_batch(2048, seed=20251226 + 2)
            orc.cycle_batch(M, S, qs[:64], fls[:64], fss[:64], cores)
            done, t1 = 0, time.perf_counter()
            while time.perf_counter() - t1 < args.cpu_seconds:
                orc.cycle_batch(M, S, qs, fls, fss, cores)
                done += qs.shape[0]
            cdt = time.perf_counter() - t1
            t2 = time.perf_counter()
            orc.cycle_batch(M, S, qs[:512], fls[:512], fss[:512], 1)
            single = 512 / (time.perf_counter() - t2)
            line["cpu_baseline"] = {
                "value": done / cdt,
                "unit": "cycles/s",
                "cores": cores,
                "kind": "port",
                "sample": f"{done} cycles of the same workload (2048-instance seeded batches repeated for {cdt:.1f} s), "
                          f"oracle/dwbc_oracle.c -O3 OpenMP over instances; single thread: {single:.0f} cycles/s",
            }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
