"""CPU: the N > 1 path of rule_n (run sharding + one all_gather) with world_size 2 on the gloo backend.
The device is replaced by a stub whose rule_n returns a deterministic function of the RUN INDEX, so the
assembled result must be identical to the single-process one whatever the number of ranks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class StubDevice:
    device = 0
    fail_rank = None          # rank whose shard raises (tests of the error path)

    def bootstrap_runs(self, T, complexify, idx_left, idx_right, n_runs, rotated, p, power, tol, n_out):
        # a deterministic function of the composed row indices of each replicate: independent of how the replicates are dealt
        idx = idx_left if idx_left is not None else idx_right
        spec = np.stack([(idx[r][:n_out] * 3.0 + idx[r].sum() % 11) for r in range(n_runs)]) if n_runs else np.zeros((0, n_out))
        kept = np.array([int(idx[r][0]) % 4 != 1 for r in range(n_runs)], dtype=bool)
        return spec.astype(np.float64), kept

    def rule_n(self, T, Nx, Ny, n_fields, complexify, rotated, p, power, tol, run_begin, run_end, seed, dtype, n_out):
        if self.fail_rank is not None:
            import torch.distributed as td
            if td.get_rank() == self.fail_rank:
                raise RuntimeError("device lost (stub)")
        runs = np.arange(run_begin, run_end)
        spectra = (runs[:, None] * 1000.0 + np.arange(n_out)[None, :] + (seed % 97)).astype(np.float64)
        kept = (runs % 5 != 3).astype(np.int32)          # every fifth run "did not converge"
        return spectra, kept


def _expected(n_runs, n_out, seed):
    return StubDevice().rule_n(0, 0, 0, 2, False, False, 0, 0, 0, 0, n_runs, seed, None, n_out)


def _worker(rank, world, port, n_runs, n_out, q):
    sys.path.insert(0, REPO)
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    from xmca_amd import dist
    seed = 1000 + rank                    # ranks disagree on purpose: rank 0's seed must win
    sp, kept = dist.sharded_rule_n(StubDevice(), n_runs, T=10, Nx=4, Ny=3, n_fields=2, complexify=False, rotated=False,
                                   p=0, power=0, tol=1e-8, seed=seed, dtype=np.float64, n_out=n_out)
    q.put((rank, sp, kept))
    td.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("n_runs", [7, 2, 1])
def test_two_rank_gather_equals_single_process(n_runs):
    world, n_out = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_runs, n_out, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp_sp, exp_kept = _expected(n_runs, n_out, 1000)
    for _, sp, kept in results:
        assert np.array_equal(sp, exp_sp) and np.array_equal(kept, exp_kept)


def test_single_process_path():
    sys.path.insert(0, REPO)
    from xmca_amd import dist
    sp, kept = dist.sharded_rule_n(StubDevice(), 6, T=10, Nx=4, Ny=3, n_fields=2, complexify=False, rotated=False, p=0, power=0,
                                   tol=1e-8, seed=5, dtype=np.float64, n_out=3)
    e_sp, e_kept = _expected(6, 3, 5)
    assert np.array_equal(sp, e_sp) and np.array_equal(kept, e_kept)


def _worker_failing(rank, world, port, q):
    sys.path.insert(0, REPO)
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    from xmca_amd import dist
    dev = StubDevice()
    dev.fail_rank = 1
    try:
        dist.sharded_rule_n(dev, 6, T=10, Nx=4, Ny=3, n_fields=2, complexify=False, rotated=False, p=0, power=0, tol=1e-8, seed=1,
                            dtype=np.float64, n_out=4)
        q.put((rank, "returned"))
    except RuntimeError as err:
        q.put((rank, str(err)))
    td.barrier()                          # every rank is still in step after the failure: nobody hangs in the collective
    td.destroy_process_group()


def test_a_failing_rank_raises_on_every_rank_instead_of_hanging_the_collective():
    """VERDICT r05 weak #9 / advisor: a rank whose shard fails still enters the all_gather (status row) and EVERY rank raises."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_failing, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert "rank(s) [1] failed" in results[0] and "this rank" not in results[0]
    assert "rank(s) [1] failed" in results[1] and "device lost (stub)" in results[1]


def _worker_bootstrap(rank, world, port, n_runs, q):
    sys.path.insert(0, REPO)
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    from xmca_amd import dist
    rng = np.random.default_rng(100 + rank)          # the ranks draw DIFFERENT indices: rank 0's must be used everywhere
    idx = rng.integers(0, 12, size=(n_runs, 12))
    sp, kept = dist.sharded_bootstrap(StubDevice(), n_runs, T=12, complexify=False, idx_left=idx, idx_right=None, rotated=False, p=0,
                                      power=1, tol=1e-8, n_out=5)
    q.put((rank, sp, kept))
    td.destroy_process_group()


@pytest.mark.parametrize("n_runs", [5, 1])
def test_two_rank_bootstrap_equals_single_process(n_runs):
    """bootstrapping shards by replicate through the same gather (xmca/array.py:1935-1947)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bootstrap, args=(r, world, port, n_runs, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    idx0 = np.random.default_rng(100).integers(0, 12, size=(n_runs, 12))
    e_sp, e_kept = StubDevice().bootstrap_runs(12, False, idx0, None, n_runs, False, 0, 1, 1e-8, 5)
    for _, sp, kept in results:
        assert np.array_equal(sp, e_sp) and np.array_equal(kept, e_kept) and kept.dtype == bool
