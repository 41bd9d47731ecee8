"""not-gpu: the N>1 path of bench.py with two gloo processes on CPU.

What runs is bench.py's own rank function (`bench.rank_main`: process-group set-up, slice bounds from libdwbc_amd/shard.py,
warm-up, timed loop between barriers, the final all_gather of [tau_total | wrench | status], max-over-ranks timing, the JSON
line) with DWBC_BENCH_BACKEND-style backend "gloo".  There is no GPU here, so the per-rank solver handed to it is a CPU
stand-in with the HipEngine interface whose numbers come from the oracle (test infrastructure); on a GPU box the same
function runs with the HIP engine (tests/test_gpu_parity.py::test_gpu_bench_two_ranks_gloo).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from tests import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleEngine:
    """CPU stand-in for bench.HipEngine: same attributes and methods, outputs from the oracle."""

    def __init__(self, args, rank, local_rank):
        import bench
        from oracle import orc

        self.dev = torch.device("cpu")
        self.q, self.fl, self.fs = bench.rank_inputs(args, rank)
        self.M = orc.make_model(cases.tocabi_model())
        self.S = orc.make_setup(cases.CONTACTS_2, cases.TASKS_2LEVEL, cases.TAU_LIM)
        self.orc = orc
        B = args.batch
        self.tau = torch.zeros((B, 3, 33), dtype=torch.float64)
        self.wrench = torch.zeros((B, 12), dtype=torch.float64)
        self.status = torch.zeros((B,), dtype=torch.int32)

    def solve(self):
        tau, wr, st, _ = self.orc.cycle_batch(self.M, self.S, self.q, self.fl, self.fs, 1)
        self.tau.copy_(torch.from_numpy(tau))
        self.wrench.copy_(torch.from_numpy(np.ascontiguousarray(wr[:, :12])))
        self.status.copy_(torch.from_numpy(st.astype(np.int32)))

    def synchronize(self):
        pass

    def kernel_ms(self, steps):
        return 0.0

    def info(self):
        return dict(kernel="oracle stand-in", threads=0, lds=0)


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import bench

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    args = bench.parse_args(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--batch", "19", "--no-cpu-baseline"])
    line, gathered = bench.rank_main(args, rank, rank, world, "gloo", engine_factory=OracleEngine)
    if rank == 0:
        np.save(os.path.join(outdir, "gathered.npy"), gathered.numpy())
        json.dump(line, open(os.path.join(outdir, "line.json"), "w"))
    else:
        assert line is None


def test_bench_rank_function_two_gloo_ranks_matches_single_solve(tmp_path):
    import bench
    from libdwbc_amd import shard
    from oracle import orc

    world, B = 2, 19
    assert shard.shard_range(37, 0, 2) == (0, 19) and shard.shard_range(37, 1, 2) == (19, 37)  # ragged slices stay contiguous
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "gathered.npy")
    line = json.load(open(tmp_path / "line.json"))
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 2
    assert line["config"]["batch_per_gpu"] == B and line["config"]["collective_backend"] == "gloo"
    # the global batch is the concatenation of the ranks' seeded batches: one solve of it must equal the gathered rows
    args = bench.parse_args(["--batch", str(B)])
    parts = [bench.rank_inputs(args, r) for r in range(world)]
    q, fl, fs = (np.concatenate([p[i] for p in parts]) for i in range(3))
    M = orc.make_model(cases.tocabi_model())
    S = orc.make_setup(cases.CONTACTS_2, cases.TASKS_2LEVEL, cases.TAU_LIM)
    tau, wr, st, _ = orc.cycle_batch(M, S, q, fl, fs, 1)
    ref = shard.pack_outputs(tau, wr[:, :12], st)
    assert got.shape == ref.shape == (world * B, 46)
    assert np.abs(got - ref).max() < 1e-12
    assert line["value"] > 0 and abs(line["value"] - world * B * 2 / (line["ms_per_step"] * 2e-3)) < 1e-6 * line["value"]


def _worker_forced(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import bench

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["DWBC_BENCH_FORCE_COLLECTIVE"] = "1"
    args = bench.parse_args(["--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "7", "--no-cpu-baseline"])
    line, gathered = bench.rank_main(args, 0, 0, 1, "gloo", engine_factory=OracleEngine)
    json.dump(dict(line=line, rows=int(gathered.shape[0])), open(os.path.join(outdir, "forced.json"), "w"))


def test_forced_one_rank_collective_runs_the_gather_branch(tmp_path):
    """DWBC_BENCH_FORCE_COLLECTIVE=1: a one-rank job still builds its process group and sends the final rows through
    all_gather_into_tensor (the switch the GPU suite uses to execute the RCCL branch on a one-GPU box)"""
    mp.spawn(_worker_forced, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    got = json.load(open(tmp_path / "forced.json"))
    assert got["rows"] == 7 and got["line"]["n_gpus"] == 1
    assert got["line"]["config"]["forced_one_rank_gather_matches_local"] is True


def test_gather_packed_handles_ragged_slices():
    """shard.gather_packed pads ragged per-rank slices; checked here on one process with a fake dist"""
    from libdwbc_amd import shard

    class FakeDist:
        def __init__(self, parts):
            self.parts = parts

        def all_gather_into_tensor(self, out, mine):
            mx = mine.shape[0]
            for r, p in enumerate(self.parts):
                out[r * mx : r * mx + p.shape[0]] = p
                out[r * mx + p.shape[0] : (r + 1) * mx] = 0

    parts = [torch.arange(19 * 46, dtype=torch.float64).reshape(19, 46), 1000 + torch.arange(18 * 46, dtype=torch.float64).reshape(18, 46)]
    full = shard.gather_packed(parts[0], FakeDist(parts), 2, [19, 18])
    assert full.shape == (37, 46)
    assert torch.equal(full[:19], parts[0]) and torch.equal(full[19:], parts[1])


def test_bench_launcher_refuses_world_size_mismatch():
    """`--gpus N` must agree with WORLD_SIZE when a launcher set it (the line would otherwise misreport n_gpus)"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
