import ctypes as C, sys, json, os
import numpy as np, torch
sys.path.insert(0, os.getcwd())
L = C.CDLL("halo2-dynamic-sha256_amd/libhsw.so")
L.hsw_engine_create_ex.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
L.hsw_witness_blocks.argtypes = [C.c_void_p] * 3 + [C.c_size_t, C.c_uint64] + [C.c_void_p] * 3 + [C.c_size_t, C.c_void_p, C.c_uint32]
L.hsw_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
L.hsw_set_timing.argtypes = [C.c_void_p, C.c_int]
h = C.c_void_p(); assert L.hsw_engine_create_ex(0, None, 8, 2, 0, C.byref(h)) == 0
L.hsw_set_timing(h, 1)
n = 4096
rng = np.random.default_rng(0xC3)
blocks = torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8)).cuda()
pre = torch.from_numpy(np.tile(np.array([0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19], dtype=np.uint32).view(np.int32), (n, 1))).cuda()
dense = torch.zeros((2, 2060 * n, 4), dtype=torch.int64, device="cuda")
spread = torch.zeros((2, 2060 * n, 4), dtype=torch.int64, device="cuda")
nxt = torch.empty((n, 8), dtype=torch.int32, device="cuda")
def run(gate):
    ts = []
    for i in range(12):
        assert L.hsw_witness_blocks(h, blocks.data_ptr(), pre.data_ptr(), n, 0, gate.data_ptr(), dense.data_ptr(), spread.data_ptr(), 2060 * n, nxt.data_ptr(), 0) == 0
        ms = C.c_float(); L.hsw_last_kernel_ms(h, C.byref(ms)); ts.append(ms.value)
    return float(np.median(ts[2:]))
tot = n * 66308 * 32                      # bytes of the gate stream (chip columns are separate buffers)
bufs = []
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for k in range(NB):                      # 24 x 8.69 GB = 209 GB of the 288 GB: a map of the card by allocation order
    g = torch.empty((n * 66308, 4), dtype=torch.int64, device="cuda")
    bufs.append(g)
t1 = [run(g) for g in bufs]
t2 = [run(g) for g in bufs]
for k, g in enumerate(bufs):
    print("buffer %2d at 0x%x: %.4f  %.4f ms" % (k, g.data_ptr(), t1[k], t2[k]), flush=True)
L.hsw_fill_calibrate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_float)]
def fill(ptr, nbytes):
    ts = []
    for _ in range(7):
        ms = C.c_float(); assert L.hsw_fill_calibrate(h, ptr, nbytes, C.byref(ms)) == 0; ts.append(ms.value)
    return nbytes / 1e6 / float(np.median(ts[2:]))
print("fill GB/s per buffer:", " ".join("%.0f" % fill(g.data_ptr(), tot) for g in bufs))
# without the chip columns (HSW_SKIP_CHIP): is it the gate stream's placement alone?
def run_nochip(gate):
    ts = []
    for i in range(10):
        assert L.hsw_witness_blocks(h, blocks.data_ptr(), pre.data_ptr(), n, 0, gate.data_ptr(), None, None, 2060 * n, nxt.data_ptr(), 4) == 0
        ms = C.c_float(); L.hsw_last_kernel_ms(h, C.byref(ms)); ts.append(ms.value)
    return float(np.median(ts[2:]))
print("no chip columns:", " ".join("%.4f" % run_nochip(g) for g in bufs))

# the chip columns somewhere else: inside buffer j (its first 2 x 540 MB are scratch then); rows = gate buffer k
def run_with_chip(gate, host):
    base = host.data_ptr()
    cd, cs = base, base + 2 * 2060 * n * 32 + (1 << 21)
    ts = []
    for i in range(10):
        assert L.hsw_witness_blocks(h, blocks.data_ptr(), pre.data_ptr(), n, 0, gate.data_ptr(), cd, cs, 2060 * n, nxt.data_ptr(), 0) == 0
        ms = C.c_float(); L.hsw_last_kernel_ms(h, C.byref(ms)); ts.append(ms.value)
    return float(np.median(ts[2:]))
hosts = [0, 5, 9, 11, 14, 17, 19, 23]
print("chip columns inside buffer j ->      " + " ".join("%6d" % j for j in hosts))
for k in (1, 4, 8, 10, 12, 16, 18, 22):
    print("gate = buffer %2d:                    " % k + " ".join("%6.3f" % (run_with_chip(bufs[k], bufs[j]) if j != k else float("nan")) for j in hosts), flush=True)
