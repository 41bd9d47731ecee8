"""Batch sharding over ranks (one process per GPU).  Instances are independent, so a rank owns a contiguous slice and
there is no exchange during the solve; the only collective is the final gather of [tau_total | wrench | status]."""
import numpy as np


def shard_range(total, rank, world):
    """Contiguous, balanced slice [lo, hi) of `total` instances for `rank`."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_outputs(tau, wrench, status):
    """(B,3,m), (B,12), (B,) -> (B, m+13) float64 rows [tau_total | wrench | status]."""
    tau = np.asarray(tau)
    return np.concatenate([tau.sum(axis=1), np.asarray(wrench), np.asarray(status, dtype=np.float64)[:, None]], axis=1)


def pack_outputs_torch(tau, wrench, status):
    """the same packing on torch tensors (any device): (B,3,m), (B,12), (B,) -> (B, m+13) float64"""
    import torch

    return torch.cat([tau.sum(dim=1), wrench, status.to(torch.float64)[:, None]], dim=1).contiguous()


def gather_packed(packed, dist, world, sizes):
    """all_gather of per-rank packed rows (torch tensor, possibly ragged over ranks) -> (sum sizes, cols)."""
    import torch

    cols = packed.shape[1]
    mx = max(sizes)
    pad = torch.zeros((mx, cols), dtype=packed.dtype, device=packed.device)
    pad[: packed.shape[0]] = packed
    out = torch.empty((world * mx, cols), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * mx : r * mx + sizes[r]] for r in range(world)], dim=0)
