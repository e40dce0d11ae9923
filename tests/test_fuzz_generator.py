"""The fuzzer's digest generator (tests/fuzz_parity.py) only draws cases the reference's asserts admit
(lib.rs:57-59, 89-90), and the oracle's whole-digest path agrees with hashlib on all of them -- CPU only."""
import hashlib

import numpy as np


def test_generated_digest_cases_are_admissible_and_hash_correctly(oracle):
    from tests.fuzz_parity import Fuzzer
    f = Fuzzer.__new__(Fuzzer)                     # the generator needs only the random stream
    f.rng = np.random.default_rng(99)
    seen_pre = seen_edge = 0
    for _ in range(40):
        nd = int(f.rng.integers(1, 4))
        sizes, msgs, pres = f._random_digests(nd, [1, 2, 3, 4], equal=bool(f.rng.integers(0, 2)))
        assert all(s % 64 == 0 and p % 64 == 0 for s, p in zip(sizes, pres))
        ref = oracle.digest_cells(msgs, sizes, pres, bool(f.rng.integers(0, 2)))
        for m, d in zip(msgs, ref["digests"]):
            assert d == hashlib.sha256(m).digest()
        seen_pre += sum(1 for p in pres if p)
        seen_edge += sum(1 for m in msgs if (len(m) + 9) % 64 in (0, 1))
    assert seen_pre >= 5 and seen_edge >= 1
