"""include/hsw.h is a C header: a plain C99 program must compile against it
(CPU) and, on the GPU box, run the reference's test_sha256_correct1 flow."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "examples", "digest_abc.c")
LIBDIR = os.path.join(ROOT, "halo2-dynamic-sha256_amd")


def _build(tmp_path, src=SRC):
    exe = str(tmp_path / os.path.splitext(os.path.basename(src))[0])
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"), src,
           "-L" + LIBDIR, "-lhsw", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_valid_c99_and_example_links(tmp_path):
    exe = _build(tmp_path)
    # without a GPU the program must fail cleanly with the ABI's status text, not crash
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 1 and "no usable HIP device" in r.stderr


@pytest.mark.gpu
def test_c_example_runs_reference_flow(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad" in r.stdout
    assert "first gate row: [0, 128, 1, 128]" in r.stdout and r.stdout.strip().endswith("ok")


WHOLE = os.path.join(ROOT, "examples", "whole_region.c")


def test_whole_region_example_links(tmp_path):
    _build(tmp_path, WHOLE)


@pytest.mark.gpu
def test_whole_region_example_runs_the_bench_circuit(tmp_path):
    """benches/digest.rs:103-129 as 9 advice columns + lookup column, from plain C."""
    import hashlib
    exe = _build(tmp_path, WHOLE)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "advice columns: 9 x 131063 rows; gate cells 1116315, lookup cells 53059" in r.stdout
    assert hashlib.sha256(bytes([1] * 56)).hexdigest() in r.stdout and r.stdout.strip().endswith("ok")
    assert " constraints, 0 violations" in r.stdout


RESULT = os.path.join(ROOT, "examples", "assigned_hash_result.c")


def test_assigned_hash_result_example_links(tmp_path):
    _build(tmp_path, RESULT)


@pytest.mark.gpu
def test_assigned_hash_result_cells_through_the_c_abi(tmp_path):
    """What the Rust shim's digest() must return -- AssignedHashResult { input_len, input_bytes, output_bytes }
    (lib.rs:31-36) -- resolved to (column, row) of the advice image for the TestCircuit (3 columns) and the
    bench circuit (9 columns), and read back there, from plain C."""
    import hashlib
    exe = _build(tmp_path, RESULT)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert hashlib.sha256(b"abc").hexdigest() in r.stdout and hashlib.sha256(b"").hexdigest() in r.stdout
    assert hashlib.sha256(bytes([1] * 56)).hexdigest() in r.stdout and r.stdout.strip().endswith("ok")
    assert "TestCircuit digest 0: input_len at (0, 0), input_bytes from (0, 46)" in r.stdout
