"""not-gpu: the N>1 path (contiguous batch slices, one final gather) with two gloo processes on CPU.
The per-rank "solver" here is the oracle (test infrastructure) standing in for the GPU kernel; what is tested is the
sharding + gather logic of libdwbc_amd/shard.py that bench.py uses with RCCL."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import cases


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from libdwbc_amd import shard
    from oracle import orc

    q, fl, fs = cases.synth_batch(total, seed=77)
    lo, hi = shard.shard_range(total, rank, world)
    M = orc.make_model(cases.tocabi_model())
    S = orc.make_setup(cases.CONTACTS_2, cases.TASKS_2LEVEL, cases.TAU_LIM)
    tau, wr, st, _ = orc.cycle_batch(M, S, q[lo:hi], fl[lo:hi], fs[lo:hi], 1)
    packed = torch.from_numpy(shard.pack_outputs(tau, wr[:, :12], st))
    sizes = [shard.shard_range(total, r, world)[1] - shard.shard_range(total, r, world)[0] for r in range(world)]
    full = shard.gather_packed(packed, dist, world, sizes)
    if rank == 0:
        np.save(os.path.join(outdir, "gathered.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_solve_matches_single(tmp_path):
    from libdwbc_amd import shard
    from oracle import orc

    total, world = 37, 2  # ragged on purpose
    assert shard.shard_range(total, 0, world) == (0, 19) and shard.shard_range(total, 1, world) == (19, 37)
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "gathered.npy")
    q, fl, fs = cases.synth_batch(total, seed=77)
    M = orc.make_model(cases.tocabi_model())
    S = orc.make_setup(cases.CONTACTS_2, cases.TASKS_2LEVEL, cases.TAU_LIM)
    tau, wr, st, _ = orc.cycle_batch(M, S, q, fl, fs, 1)
    ref = shard.pack_outputs(tau, wr[:, :12], st)
    assert got.shape == ref.shape == (total, 46)
    assert np.abs(got - ref).max() < 1e-12
