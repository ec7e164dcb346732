import os
import sys


def _cap_host_threads():
    """BLAS/OpenMP threads = the CPUs this container may actually use.  The GPU boxes show 256 cores but run under a cgroup
    quota of 16: a BLAS call with one spinning thread per core exhausts the quota and the kernel then stalls every thread of
    the process - including the one waiting for the device - for 50-80 ms at a time (measured: nr_throttled in cpu.stat)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(parts[0]) // int(parts[1])))
            else:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    per = int(f.read())
                if int(parts[0]) > 0:
                    n = min(n, max(1, int(parts[0]) // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    for v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(v, str(n))
    return n


HOST_THREADS = _cap_host_threads()          # before numpy / torch load their thread pools

import numpy as np   # noqa: E402
import pytest        # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver's GPU tier)")


@pytest.fixture(scope="session")
def hip():
    """Process-wide device handle; GPU tests fail loudly (no skip, no fallback) when no device is visible."""
    from xmca_amd import _hip
    return _hip.default_handle(0)


def align_modes(V, Vref):
    """per-mode phase/sign alignment: returns V * conj(phase) with phase_m = <Vref_m, V_m> / |.|"""
    ph = np.sum(np.conj(Vref) * V, axis=0)
    ph = np.where(np.abs(ph) > 0, ph / np.abs(ph), 1.0)
    return V / ph, ph
