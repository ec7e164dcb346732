"""Run-sharding of the Rule-N surrogate loop over the ranks of a torch.distributed job.

Surrogate runs are independent (xmca/array.py:1753-1765), so rank r of W processes the contiguous block
[r*n/W, (r+1)*n/W) of run indices on its own GPU - no data-path collective - and the per-run spectra
(<= 25 x 5000 float64 = 1 MB per rank at the largest configuration) are combined with ONE all_gather
(RCCL over xGMI when the backend is "nccl", gloo on CPU in the tests).  The device generator is keyed by
(seed, run, side), so the result does not depend on the number of ranks.

Two transports for that one collective:
  * a torch.distributed process group, when the caller has initialised one (torchrun; `bench.py`);
  * the library's own RCCL communicator (C ABI `xmca_comm_*`, `xmca_rule_n_sharded`: include/xmca_hip.h), for callers without
    torch: `init_native(dev)` reads RANK / WORLD_SIZE, rank 0 makes the ncclUniqueId and passes it through a file
    (`XMCA_COMM_ID_FILE`, default under /dev/shm keyed by MASTER_PORT: one node, as SURVEY 8(e)), and
    `sharded_rule_n(..., comm=that)` then runs shard + gather inside the library.
"""
import os
import time

import numpy as np


def _dist():
    try:
        import torch.distributed as td
    except Exception:
        return None
    if td.is_available() and td.is_initialized():
        return td
    return None


def _giveups():
    try:
        from . import _hip
        return int(_hip.load_library().xmca_persistent_giveups())
    except Exception:                                             # noqa: BLE001  (a library without the counter)
        return 0


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


def init_native(dev, rank=None, world=None, id_file=None, timeout=120.0):
    """Native RCCL communicator over the ranks of a launcher WITHOUT torch.distributed: rank 0 writes the 128-byte
    ncclUniqueId to `id_file` (atomically), the others wait for it, all call ncclCommInitRank.  Returns `_hip.Comm`.

    The rendezvous file is keyed per LAUNCH (advisor, round 5: under torchrun the run id defaults to 'none', so a file left by a
    run that died was picked up - at once - by the next run's ranks, which then hung in ncclCommInitRank on a stale id):
    `XMCA_COMM_ID_FILE`, else /dev/shm/xmca_comm_id_<uid>_<MASTER_PORT>_<run id>_<restart count>_<launcher pid>.  Readers also
    reject a file older than their own process (a leftover of an earlier launch with the same key), rank 0 removes any
    pre-existing file before it publishes, and the file is created exclusively (O_EXCL | O_NOFOLLOW, mode 0600) under a
    temporary name and renamed into place."""
    from . import _hip
    rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
    if id_file is None:
        id_file = os.environ.get("XMCA_COMM_ID_FILE") or "/dev/shm/xmca_comm_id_%d_%s_%s_%s_%d" % (
            os.getuid(), os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
            os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), os.getppid())
    started = _process_start_time()
    if rank == 0:
        try:
            os.remove(id_file)                    # a leftover with this key is never a live rendezvous: rank 0 has not published yet
        except OSError:
            pass
        uid = _hip.comm_unique_id()
        tmp = id_file + ".tmp%d" % os.getpid()
        try:
            os.remove(tmp)
        except OSError:
            pass
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(uid)
        os.replace(tmp, id_file)
    else:
        t0 = time.time()
        uid = None
        while uid is None:
            try:
                fd = os.open(id_file, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
                with os.fdopen(fd, "rb") as f:
                    st = os.fstat(f.fileno())
                    data = f.read()
                # (written by rank 0 of THIS launch: not older than this process, owned by this user)
                if len(data) == _hip.COMM_ID_BYTES and st.st_mtime >= started - 1.0 and st.st_uid == os.getuid():
                    uid = data
            except OSError:
                pass
            if uid is None:
                if time.time() - t0 > timeout:
                    raise TimeoutError("no fresh RCCL unique id at %s after %.0f s" % (id_file, timeout))
                time.sleep(0.01)
    comm = _hip.Comm(dev, uid, rank, world)          # collective: returns once every rank has joined
    if rank == 0 and world > 0:
        try:
            os.remove(id_file)
        except OSError:
            pass
    return comm


def _process_start_time():
    """wall-clock time this process started (seconds since the epoch); falls back to 'now' where /proc is not available"""
    try:
        with open("/proc/self/stat") as f:
            ticks = float(f.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/uptime") as f:
            up = float(f.read().split()[0])
        return time.time() - up + ticks / os.sysconf("SC_CLK_TCK")
    except Exception:                                             # noqa: BLE001
        return time.time()


def sharded_rule_n(dev, n_runs, *, T, Nx, Ny, n_fields, complexify, rotated, p, power, tol, seed, dtype, n_out, comm=None):
    """Returns (spectra [n_runs x n_out], kept [n_runs]) assembled on every rank.

    `comm`: a native communicator (`init_native`) - shard, seed broadcast and the all-gather then run inside the library
    (xmca_rule_n_sharded); otherwise the torch.distributed group when one is initialised; otherwise a single rank."""
    if comm is not None:
        return dev.rule_n_sharded(comm, n_runs, T, Nx, Ny, n_fields, complexify, rotated, p, power, tol, seed, dtype, n_out)
    td = _dist()
    rank, world = rank_world()
    seed = broadcast_seed(seed, dev)
    begin, end = shard_range(n_runs, rank, world)
    g0 = _giveups()
    # A rank whose own runs fail STILL enters the all_gather - with a status row - so that no rank is left waiting in the
    # collective; afterwards every rank raises (VERDICT r05 / advisor: raising here left the other ranks in all_gather for good).
    failure = None
    spectra = kept = None
    try:
        spectra, kept = dev.rule_n(T, Nx, Ny, n_fields, complexify, rotated, p, power, tol, begin, end, seed, dtype, n_out)
    except Exception as err:                                      # noqa: BLE001  (reported on every rank below)
        if td is None:
            raise
        failure = err
    if failure is None and _giveups() != g0:
        # a persistent reduction of this rank ran out of its bounded spins (another PROCESS held compute units of this GPU) and was
        # repeated launch by launch: same spectra to rounding, but another summation order - not the bits a rank on a GPU of its
        # own produces ("bit reproducibility")
        import warnings
        warnings.warn("xmca_amd: %d persistent launch(es) of rank %d were repeated on the launch-per-column path; the spectra of this "
                      "rank agree with a single-rank run to rounding, not bit for bit" % (_giveups() - g0, rank), RuntimeWarning)
    if td is None:
        return spectra, kept
    # (a one-rank group still runs the collective: the same code path whatever the world size)
    return _gather_runs(td, dev, n_runs, n_out, rank, world, spectra, kept, failure, "rule_n")


def _gather_runs(td, dev, n_runs, n_out, rank, world, values, kept, failure, what):
    """ONE all_gather of every rank's block of runs: `cap` rows of (n_out values, kept) and a status row (0 = fine).  Raises on
    EVERY rank when any rank's shard failed (the failing rank chains its own exception)."""
    import torch
    begin, end = shard_range(n_runs, rank, world)
    cdev = _comm_device(td, dev)
    cap = -(-n_runs // world)                      # largest shard
    local = np.zeros((cap + 1, n_out + 1))
    if failure is None:
        local[:end - begin, :n_out] = values
        local[:end - begin, n_out] = kept
    else:
        local[cap, 0] = 1.0
    mine = torch.from_numpy(local).to(cdev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    td.all_gather(parts, mine)
    parts = [q.cpu().numpy() for q in parts]
    failed = [r for r in range(world) if parts[r][cap, 0] != 0.0]
    if failed:
        msg = "xmca_amd: the %s shard of rank(s) %s failed; no rank returns a result" % (what, failed)
        if failure is not None:
            raise RuntimeError(msg + " (this rank: %s)" % failure) from failure
        raise RuntimeError(msg)
    rows = []
    for r in range(world):
        b, e = shard_range(n_runs, r, world)
        rows.append(parts[r][:e - b])
    full = np.concatenate(rows, axis=0) if rows else np.zeros((0, n_out + 1))
    return np.ascontiguousarray(full[:, :n_out]), full[:, n_out].astype(np.int32)


def broadcast_array(a, dev=None):
    """`a` of rank 0 on every rank (int64 / float64 arrays of equal shape on all ranks); the array itself without a group."""
    td = _dist()
    if td is None:
        return a
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a)).to(_comm_device(td, dev))
    td.broadcast(t, src=0)
    return t.cpu().numpy()


def sharded_bootstrap(dev, n_runs, *, T, complexify, idx_left, idx_right, rotated, p, power, tol, n_out):
    """Bootstrap replicates sharded like the Rule-N runs (xmca/array.py:1935-1947 is the loop; the replicates are independent once
    the row indices are COMPOSED on the host): rank r runs the contiguous block [r*n/W, (r+1)*n/W) of replicates on its own GPU
    (which holds the same fields), ONE all_gather of (n_out + 1) float64 per replicate.  `idx_*`: (n_runs, T) composed row
    indices or None; rank 0's draws are used on every rank (they come from numpy's GLOBAL generator, which the ranks need not
    share).  Returns (spectra [n_runs x n_out], kept [n_runs] bool) on every rank; any rank's failure raises on all of them."""
    td = _dist()
    rank, world = rank_world()
    if td is not None:
        if idx_left is not None:
            idx_left = broadcast_array(np.ascontiguousarray(idx_left, dtype=np.int64), dev)
        if idx_right is not None:
            idx_right = broadcast_array(np.ascontiguousarray(idx_right, dtype=np.int64), dev)
    begin, end = shard_range(n_runs, rank, world)
    failure = None
    spec = kept = None
    try:
        if end > begin:
            spec, kept = dev.bootstrap_runs(T, complexify, None if idx_left is None else idx_left[begin:end],
                                            None if idx_right is None else idx_right[begin:end], end - begin, rotated, p, power, tol, n_out)
        else:
            spec, kept = np.zeros((0, n_out)), np.zeros(0, dtype=bool)
    except Exception as err:                                      # noqa: BLE001  (reported on every rank by _gather_runs)
        if td is None:
            raise
        failure = err
    if td is None:
        return spec, np.asarray(kept, dtype=bool)
    full, k = _gather_runs(td, dev, n_runs, n_out, rank, world, spec, kept, failure, "bootstrapping")
    return full, k.astype(bool)
