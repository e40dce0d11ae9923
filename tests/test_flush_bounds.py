"""csrc/hsw_flush_bounds.hpp on the CPU: the integer logic that decides whether a flush of the write-out tile (or
one of its rows) lies wholly on one side of the FlexGate column breaks of its block, and by how many cells it is
then shifted.  Round 2's placement bug was a wrap below zero in exactly this arithmetic; it
needed a GPU and a lucky layout to show.  The header is plain host + device code, so the property -- every cell
such a flush may write lands where the cell-by-cell placement puts it -- is checked here over random geometries,
and the pre-fix lower bound is shown to fail the same check."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("flush_bounds") / "flush_bounds_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "halo2-dynamic-sha256_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "flush_bounds_check.cpp"), "-o", exe], check=True)
    return exe


@pytest.mark.parametrize("seed", [1, 2, 77031])
def test_whole_flushes_and_rows_are_shifted_by_the_gaps_they_passed(checker, seed):
    r = subprocess.run([checker, "200000", str(seed)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert " 0 mismatches" in r.stdout


def test_the_pre_fix_lower_bound_fails_the_same_check(checker):
    r = subprocess.run([checker, "200000", "1", "wrap"], capture_output=True, text=True)
    assert r.returncode == 1 and "MISMATCH" in r.stdout and "cell_base=0 fl=0" in r.stdout
