import importlib
import os
import sys

import pytest

# glibc writes its fatal messages (heap corruption, double free) to the controlling terminal unless told
# otherwise: a GPU-box run has no use for that -- keep them in the captured stderr.
os.environ.setdefault("LIBC_FATAL_STDERR_", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hsw():
    """The product package (directory name has a hyphen -> importlib)."""
    return importlib.import_module("halo2-dynamic-sha256_amd")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle -- the checker, never the thing under test."""
    from oracle import oracle as O
    O.build()
    return O


# Which expansion kernel the engines of engine_factory use for tiny batches: -1 = the product default (the
# small-batch kernel of hsw_small.hpp up to 32 blocks), 0 = always hsw_expand_kernel.  The parity modules
# run every test both ways (fixture kernel_choice) so that neither kernel loses its small-case coverage.
_DEFAULT_SPLIT = [-1]


@pytest.fixture(params=["default", "main-kernel"])
def kernel_choice(request):
    _DEFAULT_SPLIT[0] = -1 if request.param == "default" else 0
    yield request.param
    _DEFAULT_SPLIT[0] = -1


@pytest.fixture(scope="session")
def engine_factory(hsw):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    engines = []

    def make(num_bits_lookup=8, num_advice_columns=2):
        e = hsw.WitnessEngine(0, num_bits_lookup, num_advice_columns)
        e.set_option("split", _DEFAULT_SPLIT[0])
        engines.append(e)
        return e

    yield make
    for e in engines:
        e.close()
