import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("HNS_TEST_DUMP"):  # diagnosis aid: Python stacks of all threads every N seconds while a test runs
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["HNS_TEST_DUMP"]), repeat=True, file=sys.stderr)


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle is test infrastructure: make sure liboracle.so (and, when the reference checkout is present,
    oracle/_ref/libhns_ref.so and libhns_refk.so = the reference's samplers and kernel bodies) exist before any test
    imports them."""
    odir = os.path.join(ROOT, "oracle")
    have_ref = os.path.exists("/root/reference/src/Utils/Stencils.hpp")
    if not os.path.exists(os.path.join(odir, "liboracle.so")) or (
        have_ref and not all(os.path.exists(os.path.join(odir, "_ref", n)) for n in ("libhns_ref.so", "libhns_refk.so"))
    ):
        subprocess.run(["make", "-C", odir], check=True, capture_output=True)
    yield


def has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False
