import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-feature-refinement_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a bounded spin-wait of the workgroup kernels that runs out is an ERROR in the tests (a rejected LM step otherwise): VERDICT r4 #7
    os.environ.setdefault("LFR_SPIN_TIMEOUT_FATAL", "1")


@pytest.fixture(scope="session")
def lfr_lib():
    from lfr_amd import build, capi
    build.build()
    return capi.lib()
