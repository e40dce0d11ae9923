"""Host C++ of libhsw under AddressSanitizer + UndefinedBehaviorSanitizer -- CPU builds only (GPU sanitizers
are not available on the pool):

* host_sanity: the pure host entry points (ABI arithmetic, tape builder, pack plan, digest padding);
* host_lifecycle: engines, gadgets, staging buffers, host deliveries and pinned-pointer checks against a
  stand-in HIP runtime (tests/cpp/hip_stub.cpp: "device" memory is heap memory, launches do nothing), so that a
  copy longer than its allocation, a stale handle or a leak is an ordinary sanitizer report."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "halo2-dynamic-sha256_amd", "csrc")
SAN = ["-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
       "-fno-omit-frame-pointer", "-Wno-option-ignored"]
KERNEL_OBJS = ("hsw_kernels.o", "hsw_frame.o", "hsw_verify.o", "hsw_expand_l1.o", "hsw_expand_l2.o", "hsw_expand_l2_m32.o", "hsw_expand_l4.o",
               "hsw_expand_l8.o", "hsw_expand_l16.o", "hsw_expand_l8_rc.o", "hsw_expand_l16_rc.o", "hsw_small_l2.o")


def _hipcc():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    return hipcc


def _compile(hipcc, src, outdir):
    obj = os.path.join(outdir, os.path.basename(src) + ".o")
    r = subprocess.run([hipcc, "--offload-arch=gfx950"] + SAN + ["-c", src, "-o", obj], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return obj


@pytest.fixture(scope="module")
def host_objects(tmp_path_factory):
    """The three host translation units of libhsw, instrumented, + the (uninstrumented) kernel objects."""
    hipcc = _hipcc()
    kernels = [os.path.join(CSRC, f) for f in KERNEL_OBJS]
    if not all(os.path.exists(k) for k in kernels):
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j8"])
    out = str(tmp_path_factory.mktemp("san"))
    objs = [_compile(hipcc, os.path.join(CSRC, f), out) for f in ("hsw_api.cpp", "hsw_api_region.cpp", "hsw_gadget.cpp", "hsw_replay.cpp", "hsw_devmem.cpp")]
    return hipcc, out, objs, kernels


def _link_and_run(hipcc, out, objs, kernels, name, leaks):
    exe = os.path.join(out, name)
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-fsanitize=address,undefined", "-Wno-option-ignored"] + objs + kernels + ["-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=%d:abort_on_error=1" % leaks, UBSAN_OPTIONS="print_stacktrace=1")
    return subprocess.run([exe], env=env, capture_output=True, text=True, timeout=600)


def test_host_code_under_asan_ubsan(host_objects):
    hipcc, out, objs, kernels = host_objects
    test = _compile(hipcc, os.path.join(ROOT, "tests", "cpp", "host_sanity.cpp"), out)
    res = _link_and_run(hipcc, out, objs + [test], kernels, "host_sanity", leaks=0)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "host sanity ok" in res.stdout


def test_engine_and_gadget_lifetimes_under_asan_with_a_stub_runtime(host_objects):
    """Engine / gadget create-use-destroy orders, every host delivery into exact-size buffers, a larger column
    image after a reset (compact staging sized once), pinned-pointer validation on every
    hsw_witness_digests call (cached translation), no leak of device / pinned memory or events."""
    hipcc, out, objs, kernels = host_objects
    extra = [_compile(hipcc, os.path.join(ROOT, "tests", "cpp", f), out) for f in ("hip_stub.cpp", "host_lifecycle.cpp")]
    res = _link_and_run(hipcc, out, objs + extra, kernels, "host_lifecycle", leaks=1)
    assert res.returncode == 0, (res.stdout + res.stderr)[-6000:]
    assert "host lifecycle ok" in res.stdout
