import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure). Built on demand with gcc."""
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("dmpp_oracle.c", "dmpp_grid_oracle.c", "dmpp_oracle_batch.c", "dmpp_oracle.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    import oracle_binding
    return oracle_binding.Oracle(so)


@pytest.fixture(scope="session")
def dm():
    """The product binding; libdmpp.so must already be built (fails loudly otherwise)."""
    import dmpp_amd
    dmpp_amd.load_library()
    return dmpp_amd


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
