"""Host C++ of libhsw (ABI arithmetic, tape builder, pack plan, digest padding)
under AddressSanitizer + UndefinedBehaviorSanitizer -- CPU build only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "halo2-dynamic-sha256_amd", "csrc")


def test_host_code_under_asan_ubsan(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    kernels = [os.path.join(CSRC, f) for f in ("hsw_kernels.o", "hsw_frame.o", "hsw_verify.o", "hsw_expand_l1.o", "hsw_expand_l2.o",
                                               "hsw_expand_l4.o", "hsw_expand_l8.o", "hsw_expand_l16.o", "hsw_expand_l8_rc.o", "hsw_expand_l16_rc.o", "hsw_small_l2.o")]
    if not all(os.path.exists(k) for k in kernels):
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j8"])
    exe = str(tmp_path / "host_sanity")
    san = ["-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer"]
    objs = []
    for src in (os.path.join(CSRC, "hsw_api.cpp"), os.path.join(CSRC, "hsw_api_region.cpp"), os.path.join(CSRC, "hsw_gadget.cpp"),
                os.path.join(ROOT, "tests", "cpp", "host_sanity.cpp")):
        obj = str(tmp_path / (os.path.basename(src) + ".o"))
        r = subprocess.run([hipcc, "--offload-arch=gfx950"] + san + ["-c", src, "-o", obj],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        objs.append(obj)
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-fsanitize=address,undefined"] + objs + kernels + ["-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host sanity ok" in out.stdout
