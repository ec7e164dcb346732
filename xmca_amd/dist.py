"""Run-sharding of the Rule-N surrogate loop over the ranks of a torch.distributed job.

Surrogate runs are independent (xmca/array.py:1753-1765), so rank r of W processes the contiguous block
[r*n/W, (r+1)*n/W) of run indices on its own GPU - no data-path collective - and the per-run spectra
(<= 25 x 5000 float64 = 1 MB per rank at the largest configuration) are combined with ONE all_gather
(RCCL over xGMI when the backend is "nccl", gloo on CPU in the tests).  The device generator is keyed by
(seed, run, side), so the result does not depend on the number of ranks.
"""
import numpy as np


def _dist():
    try:
        import torch.distributed as td
    except Exception:
        return None
    if td.is_available() and td.is_initialized():
        return td
    return None


def rank_world():
    td = _dist()
    if td is None:
        return 0, 1
    return td.get_rank(), td.get_world_size()


def shard_range(n_runs, rank, world):
    """contiguous block of run indices of `rank`; sizes differ by at most one."""
    base, rem = divmod(n_runs, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def _comm_device(td, dev):
    import torch
    if td.get_backend() == 'nccl':
        return torch.device('cuda', getattr(dev, 'device', 0))
    return torch.device('cpu')


def broadcast_seed(seed, dev=None):
    td = _dist()
    if td is None:
        return int(seed)
    import torch
    t = torch.tensor([int(seed) & (2 ** 62 - 1)], dtype=torch.int64, device=_comm_device(td, dev))
    td.broadcast(t, src=0)
    return int(t.item())


def sharded_rule_n(dev, n_runs, *, T, Nx, Ny, n_fields, complexify, rotated, p, power, tol, seed, dtype, n_out):
    """Returns (spectra [n_runs x n_out], kept [n_runs]) assembled on every rank."""
    td = _dist()
    rank, world = rank_world()
    seed = broadcast_seed(seed, dev)
    begin, end = shard_range(n_runs, rank, world)
    spectra, kept = dev.rule_n(T, Nx, Ny, n_fields, complexify, rotated, p, power, tol, begin, end, seed, dtype, n_out)
    if td is None:
        return spectra, kept
    # (a one-rank group still runs the collective: the same code path whatever the world size)
    import torch
    cdev = _comm_device(td, dev)
    cap = -(-n_runs // world)                      # largest shard
    local = np.zeros((cap, n_out + 1))
    local[:end - begin, :n_out] = spectra
    local[:end - begin, n_out] = kept
    mine = torch.from_numpy(local).to(cdev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    td.all_gather(parts, mine)
    rows = []
    for r in range(world):
        b, e = shard_range(n_runs, r, world)
        rows.append(parts[r][:e - b].cpu().numpy())
    full = np.concatenate(rows, axis=0) if rows else np.zeros((0, n_out + 1))
    return np.ascontiguousarray(full[:, :n_out]), full[:, n_out].astype(np.int32)
