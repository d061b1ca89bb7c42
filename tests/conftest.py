import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# torch bundles its own HIP runtime (same SONAME as /opt/rocm's libamdhip64.so.7).  Tests that check
# results with torch ops on the GPU need torch's copy to be the first one the process loads: when
# libsqlrs_hip.so pulled in /opt/rocm's first, torch found "No HIP GPUs" afterwards (two HSA runtimes).
try:
    import torch
    torch.cuda.is_available()
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle as an abi.Backend (test infrastructure; built on demand with g++)."""
    from oracle_backend import load_oracle
    return load_oracle()


@pytest.fixture(scope="session")
def oracle_compat():
    """Oracle with the reference's COUNT-assign quirk switched on (count.rs:22)."""
    from oracle_backend import load_oracle
    return load_oracle(compat_count_last_batch=1)


@pytest.fixture(scope="session")
def hip():
    """The HIP backend; fails loudly (no fallback) when the library or the GPU is missing."""
    import sqlrs_amd
    return sqlrs_amd.hip(0)
