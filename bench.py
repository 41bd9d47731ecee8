#!/usr/bin/env python3
"""bench.py -- HQP control cycles/s on batched TOCABI states (BASELINE.json metric), one process per GPU.

A "step" = one pass of the fused hot path (UpdateKinematics .. CalcContactRedistribute, one kernel launch) over one
batch of B synthetic TOCABI states that are already resident in HBM.  N = 1 workload = BASELINE.json configs[1]:
batch = 1024, double support, 2-level HQP (pelvis 6D, upper-body rotation), tau limit 300, fp64.
N > 1: every rank solves its own B instances (weak scaling, no data-path collective); the only collective is the
final RCCL all_gather of (tau[33], wrench[12], status) of the last step, inside the timed region
(libdwbc_amd/shard.py: slice bounds, packing, gather).

`python bench.py --gpus N` without WORLD_SIZE in the environment is its own launcher: it starts N fresh child
processes (one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) BEFORE anything in the parent touches the
GPU and exits with their status.  Under torchrun (WORLD_SIZE set) it is a rank.  DWBC_BENCH_BACKEND=gloo rehearses the
multi-rank path on a box with fewer GPUs than ranks (ranks share devices, the gather goes through host tensors).

Timing: an untimed ramp (--ramp, default 400 launches; a fresh context runs its first ~100 launches 3-5 % slow), W untimed
warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides; MAX over ranks.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F_ALG = 1.34e6          # flop per cycle, double support 2-level with tau limit (SURVEY 8d / BASELINE.md 3)
PEAK_FP64_TFLOPS = 78.6  # MI355X fp64 vector = matrix peak (public spec; MI355X_MICROARCH.md lists no fp64 row)
PEAK_FP32_TFLOPS = 157.3  # MI355X fp32 vector peak (public spec), for --dtype f32 runs
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r04c_pmc_summary.json")  # rocprofv3 --pmc passes of this command
# The governing roof is fp64 arithmetic throughput: 78.6 TFLOP/s whether issued as VALU FMAs or as MFMA f64 (same rate on
# MI355X).  The contract knows two classes of roof, "hbm" and "mfma" (= compute); this kernel belongs to the compute class, and the
# label says which pipe its fp64 work is actually issued on so that nobody reads it as a claim of matrix-core use.
ROOF_BOUND = "fp64-fma (compute roof: the contract's 'mfma' class; VALU FMAs + ~100 MFMA f64 16x16x4 tile instructions per cycle)"
ROOF_NOTE = ("compute roof = fp64 FMA throughput, 78.6 TFLOP/s public spec (vector rate = matrix rate on MI355X); algorithmic flop of the "
             "reference's dense formulas; kernel_ms = HIP-event average over max(steps, 200) back-to-back launches")


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
        wg = int(k.get("wg", 64) or 64)  # 64 threads per instance, 128 for the two-wave kernel
        if not kernel_name.startswith(k["kernel"].split("<")[0].replace("void ", "")) or int(k["grid"]) != batch * wg:
            return None
        return (pm["FETCH_SIZE"]["mean_per_launch"] + pm["WRITE_SIZE"]["mean_per_launch"]) * 1024.0
    except Exception:
        return None


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1024, help="instances per GPU")
    ap.add_argument("--ramp", type=int, default=-1, help="untimed launches before the warm-up steps (context / clock ramp); default 400 (40 beyond 4096 instances per GPU), 0 = none")
    ap.add_argument("--workload", default="ds2", choices=["ds2", "ss3", "mixed", "reduced"],
                    help="ds2 = BASELINE configs[1] (the metric's config, default); ss3 / mixed / reduced = configs[2] / [3] / [4] "
                         "(parity-test cases; measured for DESIGN.md only)")
    ap.add_argument("--no-hqp", action="store_true", help="hqp = false: plain hierarchy + closed-form redistribution (DESIGN.md only)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"], help="arithmetic type of the kernels (f32: measured for DESIGN.md only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args(argv)


def launch_ranks(n, argv):
    """`bench.py --gpus N` started by hand: N child processes, one rank per GPU.  The parent has not imported torch or
    touched the GPU (a process that has must never be re-executed on this pool), it only waits and returns the worst status."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    # poll: a rank that dies before the rendezvous would leave the others (and a parent waiting on them in order) hanging
    rc, live = 0, list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            rc = max(rc, abs(r))
            if r != 0:
                for o in live:
                    o.terminate()
    return rc


class HipEngine:
    """The product path of one rank: a Batch of B instances on the rank's GPU with inputs and outputs bound to torch tensors."""

    def __init__(self, args, rank, local_rank):
        import torch

        import libdwbc_amd as D
        from libdwbc_amd import workloads as cases  # the TOCABI set-up and input recipe live in the package (nothing under tests/)

        self.torch = torch
        self.dev = torch.device(f"cuda:{local_rank}")
        B = args.batch
        model = D.Model.from_urdf(cases.URDF)
        wbc = D.Batch(model, B, device=local_rank, dtype=args.dtype)
        for c in cases.CONTACTS_2:
            wbc.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
        wbc.add_task(0, D.TASK_LINK_6D, 0)
        wbc.add_task(1, D.TASK_LINK_ROTATION, 15)
        wl = args.workload
        self.reduced = wl == "reduced"
        self.hqp = not args.no_hqp
        if wl == "ss3":
            wbc.add_task(2, D.TASK_LINK_6D, 12)  # swing (right) foot, SURVEY 8d config 3
        if not self.reduced:
            wbc.set_torque_limit(np.array(cases.TAU_LIM))  # the reference's reduced path runs without the limit (App. C-10)
        q, flags, fstar = rank_inputs(args, rank)
        dev = self.dev
        self.tq = torch.from_numpy(q).to(dev)
        self.tf = torch.from_numpy(flags).to(dev)
        self.ts = torch.from_numpy(fstar).to(dev)
        self.tau = torch.zeros((B, 3, 33), dtype=torch.float64, device=dev)
        self.wrench = torch.zeros((B, 12), dtype=torch.float64, device=dev)
        self.status = torch.zeros((B,), dtype=torch.int32, device=dev)
        for name, t in (("in_q", self.tq), ("in_contact", self.tf), ("in_fstar", self.ts), ("tau", self.tau), ("wrench", self.wrench), ("status", self.status)):
            wbc.bind_tensor(name, t)
        self.stream = torch.cuda.current_stream(self.dev)
        wbc.set_stream(self.stream.cuda_stream)
        self.wbc = wbc

    def solve(self):
        self.wbc.solve(hqp=self.hqp, reduced=self.reduced)

    def synchronize(self):
        self.torch.cuda.synchronize(self.dev)

    def kernel_ms(self, steps):
        """kernel-only time with HIP events on the launch stream (roofline.achieved): the average over at least 200 back-to-back
        launches whatever --steps is, so that the fixed cost of starting and draining the queue (about 0.1 ms) does not read as
        kernel time in a short run and the figure agrees with rocprofv3's per-launch average"""
        torch = self.torch
        n = max(int(steps), 200)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(self.stream)
        for _ in range(n):
            self.solve()
        ev1.record(self.stream)
        torch.cuda.synchronize(self.dev)
        return ev0.elapsed_time(ev1) / n

    def info(self):
        nt, lds = self.wbc.launch_info()
        return dict(kernel=self.wbc.kernel_name(), threads=nt, lds=lds)


def rank_inputs(args, rank):
    """Seeded synthetic inputs of one rank (SURVEY 8d recipe); the global batch of an N-rank job is their concatenation."""
    from libdwbc_amd import workloads as cases

    kw = {"ss3": dict(contact_mode="L", levels=3), "mixed": dict(contact_mode="mixed")}.get(args.workload, {})
    return cases.synth_batch(args.batch, seed=20251226 + 2 + 1000 * rank, **kw)


def rank_main(args, rank, local_rank, world, backend, engine_factory=HipEngine):
    """One rank of the job: warm up, time exactly `steps` solves + the final gather between barriers, return (on rank 0) the
    JSON line and the gathered [tau_total | wrench | status] rows of all ranks.  `engine_factory` is the solver of the rank's
    slice (the HIP engine; tests/test_sharding_gloo.py passes a CPU stand-in to drive this function without a GPU)."""
    import torch

    from libdwbc_amd import shard

    dist = None
    # DWBC_BENCH_FORCE_COLLECTIVE=1: a one-rank job still creates its process group and runs the final all_gather (world = 1), so
    # that the RCCL branch of this function executes on a one-GPU box (tests/test_gpu_parity.py::test_gpu_bench_one_rank_rccl_gather)
    collective = world > 1 or os.environ.get("DWBC_BENCH_FORCE_COLLECTIVE") == "1"
    if collective:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            s_ = socket.socket()
            s_.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
            s_.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)  # every backend: the engine's stream, events and synchronisation are this device's
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        assert dist.get_world_size() == world, (dist.get_world_size(), world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    eng = engine_factory(args, rank, local_rank)
    B = args.batch
    total = world * B
    lo, hi = shard.shard_range(total, rank, world)  # this rank's contiguous slice of the global batch
    assert hi - lo == B
    sizes = [shard.shard_range(total, r, world)[1] - shard.shard_range(total, r, world)[0] for r in range(world)]
    cdev = eng.dev if backend in (None, "nccl") else torch.device("cpu")  # where collective buffers live

    def gather_final():
        if not collective:  # one rank: the outputs already sit where the caller reads them, there is nothing to exchange
            return None
        pack = shard.pack_outputs_torch(eng.tau, eng.wrench, eng.status)
        return shard.gather_packed(pack.to(cdev), dist, world, sizes)

    # untimed ramp before the W warm-up steps: a freshly created context runs its first ~100 launches 3 - 5 % slow (clocks and caches
    # settle: 20-step regions read 82.1, 80.5, 79.7, 78.8, 78.2, 77.5 us per step back to back, tools/sync_probe.py), which a short
    # --steps run would otherwise report as the kernel's rate.  Nothing of the timed region is touched: W warm-up steps, then exactly K steps.
    ramp = getattr(args, "ramp", 0)
    if ramp < 0:
        ramp = 400 if args.batch <= 4096 else 40
    for _ in range(ramp):
        eng.solve()
    eng.synchronize()
    for _ in range(args.warmup):
        eng.solve()
    gather_final()
    eng.synchronize()
    if collective:
        dist.barrier()
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.solve()
    gathered = gather_final()
    eng.synchronize()
    if collective:
        dist.barrier()
    eng.synchronize()
    dt = time.perf_counter() - t0
    if collective:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    forced_check = None
    if not collective:
        gathered = shard.pack_outputs_torch(eng.tau, eng.wrench, eng.status)  # (after the timed region: what a gather would have returned)
    elif world == 1:  # forced one-rank collective: what came back through the backend must be this rank's own rows
        forced_check = bool(torch.equal(gathered.cpu(), shard.pack_outputs_torch(eng.tau, eng.wrench, eng.status).cpu()))
    status_ok = float(eng.status.float().mean().item())
    kern_ms = eng.kernel_ms(args.steps)
    line = None
    if rank == 0:
        line = make_line(args, world, dt, kern_ms, status_ok, eng.info(), backend)
        if forced_check is not None:
            line["config"]["forced_one_rank_gather_matches_local"] = forced_check
        if args.workload == "reduced" and engine_factory is HipEngine:
            # configs[4] names a "reduced centroidal-dynamics fast path": on this kernel it is NOT one (the 39-wide sweeps are shared,
            # the reduced blocks come on top) -- the full-model kernel at the same batch and dtype is timed beside it so that nobody
            # reads the reduced rate as an acceleration
            import copy

            fa = copy.copy(args)
            fa.workload = "ds2"
            full = HipEngine(fa, rank, local_rank)
            for _ in range(args.warmup):
                full.solve()
            fms = full.kernel_ms(args.steps)
            line["config"]["full_model_same_batch"] = {
                "cycles_per_s": args.batch / (fms * 1e-3), "kernel_ms": fms, "kernel": full.info()["kernel"],
                "note": "the reduced (centroidal) path is a coverage row, not a fast path, on this design: reduced / full = "
                        + f"{(args.batch / (kern_ms * 1e-3)) / (args.batch / (fms * 1e-3)):.2f}"}
    if collective:
        dist.barrier()
        dist.destroy_process_group()
    return line, gathered


def make_line(args, world, dt, kern_ms, status_ok, info, backend):
    B, wl = args.batch, args.workload
    f_alg = {"ds2": F_ALG, "ss3": 1.37e6, "mixed": F_ALG, "reduced": F_ALG}[wl]
    workload_name = {
        "ds2": "BASELINE configs[1]: batch=1024 per GPU, TOCABI double support, 2-level HQP (pelvis 6D + upper-body rotation), tau limit 300, fp64",
        "ss3": "BASELINE configs[2]: TOCABI left single support + swing-foot task (3-level HQP), tau limit 300, fp64",
        "mixed": "BASELINE configs[3]: TOCABI mixed contact modes LR/L/R = 1/2,1/4,1/4 per instance, 2-level HQP, tau limit 300, fp64",
        "reduced": "BASELINE configs[4] in fp64: TOCABI double support through the reduced (centroidal) dynamics path, 2-level HQP, no tau limit",
    }[wl]
    value = world * B * args.steps / dt
    achieved = f_alg * B / (kern_ms * 1e-3) / 1e12 if kern_ms else None
    peak = PEAK_FP32_TFLOPS if args.dtype == "f32" else PEAK_FP64_TFLOPS
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
            "threads_per_instance": info["threads"],
            "lds_bytes_per_instance": info["lds"],
            "status_ok_fraction": status_ok,
            "collective_backend": {None: "none (one rank)", "nccl": "rccl"}.get(backend, backend),
            "untimed_ramp_launches": (lambda r: (400 if args.batch <= 4096 else 40) if r < 0 else r)(getattr(args, "ramp", 0)),
        },
        "roofline": {
            "bound": ROOF_BOUND,
            "achieved": achieved,
            "peak": peak,
            "unit": "TFLOP/s",
            "frac": achieved / peak if achieved else None,
            "traffic": hbm_traffic_per_launch(info["kernel"], B),
            "traffic_unit": "bytes per launch, from the committed rocprofv3 PMC passes of this command (FETCH_SIZE + WRITE_SIZE, "
                            + os.path.relpath(PMC_SUMMARY, ROOT) + "), not measured in this run",
            "kernel": info["kernel"],
            "kernel_ms": kern_ms,
            "flop_per_cycle": f_alg,
            "note": ROOF_NOTE,
        },
    }
    return line


def cpu_baseline(args):
    """The CPU restatement (oracle, OpenMP over instances) timed on this box's host cores on a bounded sample of the workload."""
    from oracle import orc
    from tests import cases

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
    return {
        "value": done / cdt,
        "unit": "cycles/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{done} cycles of the same workload (2048-instance seeded batches repeated for {cdt:.1f} s), "
                  f"oracle/dwbc_oracle.c -O3 OpenMP over instances; single thread: {single:.0f} cycles/s",
    }


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            return launch_ranks(args.gpus, argv)  # nothing above has touched the GPU
        world, rank, local_rank = 1, 0, 0
    else:
        world = int(os.environ["WORLD_SIZE"])
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {world}", file=sys.stderr)
        return 2
    backend = None
    if world > 1 or os.environ.get("DWBC_BENCH_FORCE_COLLECTIVE") == "1":
        # DWBC_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks.  The driver's runs use
        # the default, RCCL with one GPU per rank.
        backend = os.environ.get("DWBC_BENCH_BACKEND", "nccl")
        if backend != "nccl":
            import torch

            local_rank = local_rank % max(1, torch.cuda.device_count())
    line, _ = rank_main(args, rank, local_rank, world, backend)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and args.workload == "ds2":
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
