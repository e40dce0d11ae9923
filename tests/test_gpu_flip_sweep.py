"""Completeness of the on-device verifier (tests/flip_sweep.py): corrupting ANY single cell of a block --
gate stream, chip columns, lookup column, next state, input bytes, pre-state -- is reported; in the digest
frames the only cells whose corruption passes are the two witnesses per digest the reference's circuit itself
leaves free: the inverse witness of an is_zero whose input IS zero (row z + a*inv = 1 with a = 0) -- the
prologue's is_zero(limb1) (lib.rs:142-143) and the epilogue's is_equal of the selected round (lib.rs:296-310)."""
import pytest

pytestmark = pytest.mark.gpu


def test_every_cell_of_a_block_is_pinned(hsw):
    from tests.flip_sweep import sweep
    missed = sweep(8, 2, mont=False, internals=True)
    assert all(len(v) == 0 for v in missed.values()), {k: v[:10] for k, v in missed.items() if v}


@pytest.mark.parametrize("bits,ncols,mont,internals,limb", [(8, 2, True, True, 0), (16, 1, False, False, 3), (4, 3, True, True, 1)])
def test_every_cell_is_pinned_other_shapes_sampled(hsw, bits, ncols, mont, internals, limb):
    from tests.flip_sweep import sweep
    missed = sweep(bits, ncols, mont=mont, internals=internals, limb=limb, stride=5)
    assert all(len(v) == 0 for v in missed.values()), {k: v[:10] for k, v in missed.items() if v}


@pytest.mark.parametrize("mont,columns", [(False, None), (True, 100003)])
def test_frames_leave_only_the_is_zero_inverses_free(hsw, mont, columns):
    from tests.flip_sweep import sweep_frames
    missed = sweep_frames(sizes=(128, 64), rc=True, mont=mont, columns=columns)
    # digest 0: 60 bytes -> 2 rounds of 2 blocks: candidate 2 matches; digest 1: "abc", candidate 1 matches.
    # P_ISZ + 2 = 27 is the inverse of is_zero(limb1); 76*target + 4 + 2 the inverse of the matching is_equal.
    assert missed["frame gate"] == [(0, 27, "prologue"), (0, 76 * 2 + 6, "epilogue"),
                                    (1, 27, "prologue"), (1, 76 * 1 + 6, "epilogue")]
    assert missed["frame lookup"] == []
