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

    def rule_n(self, T, Nx, Ny, n_fields, complexify, rotated, p, power, tol, run_begin, run_end, seed, dtype, n_out):
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
