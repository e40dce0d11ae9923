"""A short fixed-seed slice of tests/fuzz_parity.py (randomised differential run, HIP path vs oracle) in
the GPU suite; the long runs are `python tests/fuzz_parity.py <seconds> <seed>`."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [20261004, 7])
def test_fuzz_slice(hsw, oracle, seed):
    from tests.fuzz_parity import Fuzzer
    f = Fuzzer(seed)
    try:
        out = f.run(iterations=30)
    finally:
        f.close()
    assert out["block_runs"] >= 10 and out["digest_runs"] >= 3, out
