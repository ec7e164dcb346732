import os
import sys

import numpy as np
import pytest

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
