#!/usr/bin/env python3
"""bench.py -- SHA-256 witness-assignment throughput on MI355X.

Metric (BASELINE.json): SHA256 compression blocks/sec (witness assign) +
achieved HBM GB/s at the k=17 circuit shape (num_bits_lookup=8,
num_advice_columns=2).

A "step" is one pass of the hot path (hsw_witness_blocks: chain seeds ->
gate-cell stream + spread-chip columns + next states) over one batch of
synthetic single-block messages that is already resident in HBM.

  N=1 workload  BASELINE.json configs[2]: 4,096 independent single-block
                (55-byte) messages, seed 0xC3 -- the configuration the
                metric's "achieved HBM GB/s" is quoted on.  The reference's own
                bench workload (benches/digest.rs: one 56-byte message, 16
                blocks, 9 advice columns; configs[0]) and configs[1] (one
                16-block message) are latency-bound launches of 38 MB: they are
                timed too, each with its own cpu_baseline and roofline
                fraction, under "extra".
  N>1           `python bench.py --gpus N` starts N ranks itself (one fresh
                process per GPU, RCCL over xGMI) unless it already runs under
                torch.distributed.run (WORLD_SIZE set).  Every rank generates
                its own 4,096 messages (weak scaling, no data-path collective:
                messages are independent).  The RCCL all-gather of witness
                columns that north_star also asks for is timed separately on a
                bounded shard ("extra.multi_gpu") -- it never enters `value`.

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6300.0  # MI355X_MICROARCH.md: ~6.0-6.3 TB/s is what a streaming kernel reaches on this part
IV = np.array([0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
               0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19], dtype=np.uint32)


def sha_pad_single_block(msgs55):
    """(n,55) message bytes -> (n,64) padded blocks (lib.rs:98-108 padding)."""
    n = msgs55.shape[0]
    blocks = np.zeros((n, 64), dtype=np.uint8)
    blocks[:, :55] = msgs55
    blocks[:, 55] = 0x80
    bitlen = 55 * 8
    blocks[:, 62] = (bitlen >> 8) & 0xFF
    blocks[:, 63] = bitlen & 0xFF
    return blocks


def host_cores():
    """Cores this process may actually use: min(affinity mask, cgroup cpu quota).
    A 1-GPU box of the pool exposes every host core in the affinity mask but
    shares them between tenants (16 per GPU): without a readable quota the
    count is capped at 16 x visible GPUs.  HSW_BENCH_CORES overrides."""
    if os.environ.get("HSW_BENCH_CORES"):
        return max(1, int(os.environ["HSW_BENCH_CORES"]))
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(period)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except Exception:
            pass
    if quota is not None:
        cores = max(1, min(cores, int(quota)))
    else:
        try:
            import torch
            ngpu = max(1, torch.cuda.device_count())
        except Exception:
            ngpu = 1
        cores = min(cores, 16 * ngpu)
    return cores


def cpu_baseline(blocks, pre, cpu_seconds_target=16.0):
    """Time the CPU oracle (kind 'port': C restatement of the reference's Rust
    path; the Rust crate itself cannot be built offline) on a bounded sample of
    the same workload: every usable host core expands the same kind of
    single-block messages into its own stream buffers (streams written, checks
    off), about `cpu_seconds_target` seconds of CPU work in total."""
    import threading
    from oracle import oracle as O
    O.build()
    L = O.lib()
    cores = host_cores()
    G, LC = O.measure_shape(8, 2)
    chunk = 64                                   # blocks per oracle call: 153 MB of cells per thread
    # calibrate the single-thread rate
    o = O.Oracle(8, 2, check=False)
    o.witness_blocks(blocks[:4], pre[:4], want_streams=False)          # warm caches / code
    t0 = time.perf_counter()
    o.witness_blocks(blocks[:32], pre[:32], want_streams=False)
    dt1 = (time.perf_counter() - t0) / 32          # compute only; the threads below also write the streams
    rounds = max(1, int(round(cpu_seconds_target / (dt1 * chunk * cores))))

    def work(i, res):
        h = L.oracle_create(8, 2, 0)
        gate = np.empty((chunk * G, 4), dtype=np.uint64)
        dense = np.empty((2, chunk * LC // 2, 4), dtype=np.uint64)
        spread = np.empty((2, chunk * LC // 2, 4), dtype=np.uint64)
        nxt = np.zeros((chunk, 8), dtype=np.uint32)
        for a in (gate, dense, spread):
            a.fill(0)                              # map the pages before the clock starts (a prover reuses its buffers)
        barrier.wait()
        t = time.perf_counter()
        for r in range(rounds):
            lo = ((i * rounds + r) * chunk) % max(1, blocks.shape[0] - chunk + 1)
            b = np.ascontiguousarray(blocks[lo:lo + chunk])
            p = np.ascontiguousarray(pre[lo:lo + chunk])
            L.oracle_set_cursor(h, 0)
            L.oracle_set_outputs(h, gate.ctypes.data, gate.shape[0], dense.ctypes.data, spread.ctypes.data,
                                 dense.shape[1], 0)
            L.oracle_witness_blocks(h, b.ctypes.data, p.ctypes.data, chunk, nxt.ctypes.data)
        res[i] = time.perf_counter() - t
        L.oracle_destroy(h)

    res = [0.0] * cores
    barrier = threading.Barrier(cores)
    th = [threading.Thread(target=work, args=(i, res)) for i in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = max(res)                                # all threads start together at the barrier
    n_blocks = cores * rounds * chunk
    return {
        "value": n_blocks / wall,
        "unit": "blocks/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d blocks (%d threads x %d rounds x %d blocks of the same 55-byte messages), "
                  "oracle/hsw_oracle.c -O3 -march=native, streams written to pre-mapped buffers, checks off; "
                  "%.1f s wall, %.1f s of CPU work; single-thread compute-only %.0f blocks/s" % (
                      n_blocks, cores, rounds, chunk, wall, sum(res), 1.0 / dt1),
        "value_1thread": 1.0 / dt1,
    }


def cpu_baseline_digest(message, max_size, range_check, cpu_seconds_target=6.0):
    """The reference's own bench workload on the CPU: ONE Sha256DynamicConfig::digest with every cell of
    the region (prologue, max_size/64 blocks, epilogue, lookup column, chip columns -- oracle_digest_cells,
    the restatement of lib.rs:71-349 under A1-A4), checks off, streams written to pre-mapped buffers.
    First on one thread -- the reference's synthesize() is single-threaded -- then one synthesis per core."""
    import ctypes as C
    import threading
    from oracle import oracle as O
    O.build()
    L = O.lib()
    nblk = max_size // 64
    G, LC = O.measure_shape(8, 2, True)
    LK = O.lookup_cells_per_block(8, 2)
    gcap = 64 + 5 * max_size + nblk * G + 76 * (nblk + 1) + 288
    lcap = 8 + 2 * max_size + nblk * LK + 64
    msg = np.frombuffer(bytes(message), dtype=np.uint8).copy()

    def make_bufs():
        b = (np.zeros((gcap, 4), dtype=np.uint64), np.zeros((lcap, 4), dtype=np.uint64),
             np.zeros((2, nblk * LC // 2 + 1, 4), dtype=np.uint64), np.zeros((2, nblk * LC // 2 + 1, 4), dtype=np.uint64))
        return b

    def synth(bufs):
        gate, lookup, dense, spread = bufs
        h = L.oracle_create(8, 2, 0)                 # a fresh Context + SpreadConfig per synthesis (lib.rs:440)
        L.oracle_set_internals(h, 1)
        L.oracle_set_outputs(h, gate.ctypes.data, gcap, dense.ctypes.data, spread.ctypes.data, dense.shape[1], 0)
        L.oracle_set_lookup_output(h, lookup.ctypes.data, lcap)
        dig = np.zeros(32, dtype=np.uint8)
        lay = O.DigestLayout()
        rc = L.oracle_digest_cells(h, msg.ctypes.data, len(msg), 0, max_size, 1 if range_check else 0,
                                   dig.ctypes.data, C.byref(lay))
        L.oracle_destroy(h)
        assert rc < 10
        return dig.tobytes()

    import hashlib
    bufs = make_bufs()
    assert synth(bufs) == hashlib.sha256(bytes(message)).digest()
    t0 = time.perf_counter()
    reps1 = 0
    while time.perf_counter() - t0 < cpu_seconds_target / 3 or reps1 < 3:
        synth(bufs)
        reps1 += 1
    dt1 = (time.perf_counter() - t0) / reps1
    cores = host_cores()
    rounds = max(2, int(round(cpu_seconds_target * 2 / 3 / dt1 / 1.0)) // max(1, cores) + 1)
    res = [0.0] * cores
    barrier = threading.Barrier(cores)

    def work(i):
        b = make_bufs()
        synth(b)
        barrier.wait()
        t = time.perf_counter()
        for _ in range(rounds):
            synth(b)
        res[i] = time.perf_counter() - t

    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = max(res)
    return {
        "value": nblk / dt1, "unit": "blocks/s", "cores": 1, "kind": "port",
        "ms_per_synthesis": dt1 * 1e3,
        "sample": "%d syntheses of the benches/digest.rs region (1 x %d-byte message, max %d B = %d blocks, range "
                  "checks %s) on ONE thread -- the reference's synthesize() is single-threaded; oracle/hsw_oracle.c "
                  "oracle_digest_cells, checks off, all streams written" % (reps1, len(message), max_size, nblk,
                                                                             "on" if range_check else "off"),
        "all_cores": {"value": cores * rounds * nblk / wall, "unit": "blocks/s", "cores": cores,
                      "ms_per_synthesis_amortised": wall / (cores * rounds) * 1e3,
                      "sample": "%d threads x %d independent syntheses" % (cores, rounds)},
    }


# ------------------------------------------------------------------ N > 1: start the ranks
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """`python bench.py --gpus N` outside torch.distributed.run: start N fresh rank processes (one per GPU)
    and relay rank 0's JSON line.  Runs BEFORE this process touches HIP (torch.cuda.device_count() does
    not initialise the GPU on this image), and never re-execs anything: the ranks are children."""
    n = args.gpus
    same_device = os.environ.get("HSW_BENCH_SAME_DEVICE") == "1"      # rehearsal on a 1-GPU box (with HSW_BENCH_BACKEND=gloo)
    import torch
    visible = torch.cuda.device_count()
    if not same_device and visible < n:
        sys.stderr.write("bench.py: --gpus %d asked for, but only %d GPU(s) visible\n" % (n, visible))
        return 2
    env0 = dict(os.environ)
    env0.setdefault("MASTER_ADDR", "127.0.0.1")
    env0.setdefault("MASTER_PORT", str(_free_port()))
    env0["WORLD_SIZE"] = str(n)
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        env = dict(env0)
        env["RANK"] = env["LOCAL_RANK"] = str(r)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode]
    deadline = time.time() + 120
    for p in procs[1:]:
        try:
            rcs.append(p.wait(timeout=max(1.0, deadline - time.time())))
        except subprocess.TimeoutExpired:
            p.kill()                                  # exactly the child we started
            rcs.append(-9)
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(rcs) if c != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed: %s\n" % bad)
        return 1
    return 0


# ------------------------------------------------------------------ extras (rank 0, outside the timed region)
def extra_representations(hsw, eng, blocks, pre, cursor0, out, n, alg_bytes, step):
    extra = {}
    # same batch with cells in Montgomery form (x * 2^256 mod p: halo2curves' in-memory Fr)
    try:
        for _ in range(2):
            eng.witness_blocks(blocks, pre, cursor0=cursor0, out=out, flags=hsw.HSW_REPR_MONTGOMERY)
        mm = []
        for _ in range(5):
            eng.witness_blocks(blocks, pre, cursor0=cursor0, out=out, flags=hsw.HSW_REPR_MONTGOMERY)
            mm.append(eng.last_kernel_ms())
        m_ms = float(np.median(mm))
        extra["montgomery_repr"] = {"kernel": eng.last_launch()["kernel"], "kernel_ms": m_ms, "blocks_per_s": n / m_ms * 1e3,
                                    "GBps": alg_bytes * n / m_ms / 1e6, "frac_of_peak": alg_bytes * n / m_ms / 1e6 / HBM_PEAK_GBS}
        try:     # the Montgomery stream checked on the device (cells reduced on load)
            reps_m = [eng.verify_blocks(blocks, pre, out, cursor0=cursor0, flags=hsw.HSW_REPR_MONTGOMERY) for _ in range(3)]
            extra["montgomery_repr"]["verify_on_device"] = {
                "violations": reps_m[-1]["violations"], "kernel_ms": float(np.median([r["kernel_ms"] for r in reps_m]))}
        except Exception as ex:
            extra["montgomery_repr"]["verify_on_device"] = {"error": repr(ex)}
        step()      # leave canonical cells in the buffers
    except Exception as ex:
        extra["montgomery_repr"] = {"error": repr(ex)}
    # 8-byte transport cells (HSW_REPR_COMPACT64): a quarter of the bytes, so no longer HBM-bound
    try:
        oc = eng.alloc_outputs(n, cursor0, hsw.HSW_REPR_COMPACT64)
        cm = []
        for i in range(7):
            eng.witness_blocks(blocks, pre, cursor0=cursor0, out=oc, flags=hsw.HSW_REPR_COMPACT64)
            if i >= 2:
                cm.append(eng.last_kernel_ms())
        c_ms = float(np.median(cm))
        cbytes = (alg_bytes - 128) // 4 + 128
        extra["compact64_repr"] = {"kernel": eng.last_launch()["kernel"], "kernel_ms": c_ms, "blocks_per_s": n / c_ms * 1e3,
                                   "GBps": cbytes * n / c_ms / 1e6, "bound": "instruction issue / LDS, not HBM"}
        del oc
    except Exception as ex:
        extra["compact64_repr"] = {"error": repr(ex)}
    # the product's own MockProver-style check of the batch just written, in HBM (hsw_verify_blocks)
    try:
        step()
        reps_v = [eng.verify_blocks(blocks, pre, out, cursor0=cursor0) for _ in range(3)]
        vms = float(np.median([r["kernel_ms"] for r in reps_v]))
        extra["verify_on_device"] = {"violations": reps_v[-1]["violations"], "checks": reps_v[-1]["checks"],
                                     "kernel_ms": vms, "blocks_per_s": n / vms * 1e3,
                                     "read_GBps": alg_bytes * n / vms / 1e6,
                                     "note": "every gate row, copy constraint, constant, range bound, chip cell / spread-table row and next state of all blocks"}
    except Exception as ex:
        extra["verify_on_device"] = {"error": repr(ex)}
    return extra


def extra_config1(hsw, eng, dev, local_rank, alg_bytes):
    """configs[1]: one 1 KiB-class message = 16 chained blocks (1,015 bytes)."""
    import hashlib
    import torch
    res = {}
    m = bytes(((i * 131 + 7) % 256) for i in range(1015))
    padded = bytearray(m) + b"\x80" + b"\x00" * ((64 - (len(m) + 9) % 64) % 64) + (8 * len(m)).to_bytes(8, "big")
    assert len(padded) == 1024
    b16 = torch.from_numpy(np.frombuffer(bytes(padded), dtype=np.uint8).reshape(16, 64).copy()).to(dev)
    o16 = eng.alloc_outputs(16, 0)
    for _ in range(3):
        p16 = eng.sha256_chain(b16, 1, 16)
        eng.witness_blocks(b16, p16, out=o16)
    torch.cuda.synchronize()
    reps = 50
    km = []
    for _ in range(12):                             # the expansion launch alone (HIP events on the engine's stream)
        eng.witness_blocks(b16, p16, out=o16)
        km.append(eng.last_kernel_ms())
    k_ms = float(np.median(km[2:]))
    li = eng.last_launch()
    t1 = time.perf_counter()
    for _ in range(reps):
        p16 = eng.sha256_chain(b16, 1, 16)
        eng.witness_blocks(b16, p16, out=o16)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t1) / reps
    last = o16["next_states"][15].cpu().numpy().view(np.uint32)
    assert b"".join(int(x).to_bytes(4, "big") for x in last) == hashlib.sha256(m).digest()
    res["expand_kernel"] = {"kernel": li["kernel"], "parts": li["parts"], "split": li["split"], "ms": k_ms,
                            "GBps": 16 * alg_bytes / k_ms / 1e6, "frac_of_peak": 16 * alg_bytes / k_ms / 1e6 / HBM_PEAK_GBS}
    res["device_resident_chain_plus_expand"] = {"ms_per_message": dt * 1e3, "blocks_per_s": 16 / dt,
                                                "GBps": 16 * alg_bytes / dt / 1e9,
                                                "frac_of_peak": 16 * alg_bytes / dt / 1e9 / HBM_PEAK_GBS}
    try:   # HSW_CHAINED: ONE launch, every wave derives its block's pre-state itself (no chain pre-pass)
        iv = torch.from_numpy(IV.view(np.int32).copy()).to(dev)
        CH = hsw._native.HSW_CHAINED
        for _ in range(3):
            eng.witness_blocks(b16, iv, out=o16, flags=CH)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            eng.witness_blocks(b16, iv, out=o16, flags=CH)
        torch.cuda.synchronize()
        dtc = (time.perf_counter() - t1) / reps
        kc = []
        for _ in range(8):
            eng.witness_blocks(b16, iv, out=o16, flags=CH)
            kc.append(eng.last_kernel_ms())
        last = o16["next_states"][15].cpu().numpy().view(np.uint32)
        assert b"".join(int(x).to_bytes(4, "big") for x in last) == hashlib.sha256(m).digest()
        res["device_resident_chained_single_launch"] = {
            "ms_per_message": dtc * 1e3, "blocks_per_s": 16 / dtc, "kernel_ms": float(np.median(kc)),
            "GBps": 16 * alg_bytes / dtc / 1e9, "frac_of_peak": 16 * alg_bytes / dtc / 1e9 / HBM_PEAK_GBS,
            "note": "HSW_CHAINED: d_pre_states = the message's initial state; block b's waves walk b compressions to their pre-state"}
    except Exception as ex:
        res["device_resident_chained_single_launch"] = {"error": repr(ex)}
    try:   # the same two launches captured into one HIP graph and replayed
        gs = torch.cuda.Stream()
        eng_g = hsw.WitnessEngine(local_rank, 8, 2, stream=gs)
        with torch.cuda.stream(gs):
            pg = eng_g.sha256_chain(b16, 1, 16)
            eng_g.witness_blocks(b16, pg, out=o16)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=gs):
            pg = eng_g.sha256_chain(b16, 1, 16)
            eng_g.witness_blocks(b16, pg, out=o16)
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            graph.replay()
        torch.cuda.synchronize()
        dtr = (time.perf_counter() - t1) / reps
        res["hip_graph_replay"] = {"ms_per_message": dtr * 1e3, "blocks_per_s": 16 / dtr}
        del graph
        eng_g.close()
    except Exception as ex:
        res["hip_graph_replay"] = {"error": repr(ex)}
    # the same message through the gadget front-end (Sha256DynamicConfig::digest, lib.rs:71-349):
    # host padding + chain, one expansion launch, the states back, sync
    cfg = hsw.Sha256DynamicConfig(eng, [1024] * 64, True)
    for _ in range(4):
        cfg.digest(m)
    tg = []
    for _ in range(50):
        t1 = time.perf_counter()
        r = cfg.digest(m)
        tg.append(time.perf_counter() - t1)
    dtg = float(np.median(tg))
    assert r.output_bytes == hashlib.sha256(m).digest()
    cfg.close()
    res["gadget_digest_end_to_end"] = {
        "ms_per_message": dtg * 1e3, "blocks_per_s": 16 / dtg, "ms_mean": float(np.mean(tg)) * 1e3,
        "ms_max": float(np.max(tg)) * 1e3, "slowest_iteration": int(np.argmax(tg))}
    return res


def extra_config0(hsw, local_rank, with_cpu):
    """BASELINE configs[0] = the reference's own bench circuit (benches/digest.rs:93,102-109,129): one 56-byte
    message, max 1024 B => 16 blocks synthesised, input range checks, k = 17 -- as the literal advice-column
    image of the whole region (SURVEY 8 f2 + f4, assumptions A1-A4): 9 FlexGate columns x 131,063 rows + the
    lookup column + the chip columns.  north_star's target is stated on this workload, so it carries its own
    roofline fraction and cpu_baseline."""
    import ctypes as C
    import hashlib
    res = {}
    eng_i = hsw.WitnessEngine(local_rank, 8, 2, mode=hsw._native.HSW_MODE_HALO2_INTERNALS)
    cfgw = hsw.Sha256DynamicConfig(eng_i, [1024], True, whole_digest=True)
    ncol = cfgw.set_columns((1 << 17) - 9)
    m56 = bytes([1] * 56)
    # timed through the C ABI itself (hsw_gadget_reset + hsw_gadget_digest on prebuilt arguments): the Python
    # wrapper's marshalling (~10 us per call) is plumbing, not the product
    L = eng_i.lib
    mbuf = (C.c_uint8 * 56).from_buffer_copy(m56)
    hres = hsw._native.HashResult()
    for _ in range(6):
        assert L.hsw_gadget_reset(cfgw.h) == 0 and L.hsw_gadget_digest(cfgw.h, mbuf, 56, 0, C.byref(hres)) == 0
    tw = []
    for _ in range(200):
        assert L.hsw_gadget_reset(cfgw.h) == 0
        t1 = time.perf_counter()
        rcw = L.hsw_gadget_digest(cfgw.h, mbuf, 56, 0, C.byref(hres))
        tw.append(time.perf_counter() - t1)
        assert rcw == 0
    dtw = float(np.median(tw))
    assert bytes(hres.output_bytes) == hashlib.sha256(m56).digest()
    launched = eng_i.last_launch()
    vw = cfgw.view()
    chip_cells = 2 * int(vw.num_limb_sum)
    region_bytes = (int(vw.gate_cells) + int(vw.lookup_cells) + chip_cells) * 32
    # ... and on to the host, where a CPU prover (create_proof) would read the advice columns
    hostimg = cfgw.download_region(pinned=True)
    dst = hsw._native.RegionHost(hostimg["gate"].ctypes.data, hostimg["lookup"].ctypes.data, None, None)
    tdl = []
    for _ in range(9):
        assert L.hsw_gadget_reset(cfgw.h) == 0
        t1 = time.perf_counter()
        assert L.hsw_gadget_digest(cfgw.h, mbuf, 56, 0, C.byref(hres)) == 0
        assert L.hsw_gadget_download_region(cfgw.h, C.byref(dst)) == 0
        tdl.append(time.perf_counter() - t1)
    # the same delivery in the compact transport form: 8-byte cells + the side list of the wider cells
    tdc, n_wide = [], None
    try:
        cb, n_wide = cfgw.download_region_compact()
        cdst = hsw._native.RegionCompact(cb["gate"].ctypes.data, cb["lookup"].ctypes.data, None, None,
                                         cb["wide"].ctypes.data, cb["cap"], 0)
        for _ in range(9):
            assert L.hsw_gadget_reset(cfgw.h) == 0
            t1 = time.perf_counter()
            assert L.hsw_gadget_digest(cfgw.h, mbuf, 56, 0, C.byref(hres)) == 0
            assert L.hsw_gadget_download_region_compact(cfgw.h, C.byref(cdst)) == 0
            tdc.append(time.perf_counter() - t1)
        wg = cfgw.widen(cb["gate"], 0, cb["wide"], n_wide).reshape(hostimg["gate"].shape)
        assert np.array_equal(wg, hostimg["gate"]), "compact delivery differs from the 32-byte image"
    except Exception as ex:
        tdc, n_wide = None, repr(ex)
    vrep = cfgw.verify()
    res["whole_region"] = {
        "ms_per_synthesis": dtw * 1e3, "ms_min": float(np.min(tw)) * 1e3, "ms_p90": float(np.percentile(tw, 90)) * 1e3,
        "blocks_per_s": 16 / dtw, "advice_columns": ncol,
        "gate_cells": int(vw.gate_cells), "lookup_cells": int(vw.lookup_cells), "chip_cells": chip_cells,
        "kernel": launched["kernel"], "waves_per_block": launched["parts"], "split": launched["split"], "grid": launched["grid"],
        "note": "Sha256DynamicConfig::digest of benches/digest.rs through hsw_gadget_digest: prologue + 16 blocks + "
                "epilogue written as FlexGate columns + lookup column + chip columns in HBM, states back on the host"}
    res["roofline"] = {"bound": "hbm", "achieved": region_bytes / dtw / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": region_bytes / dtw / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": region_bytes,
                       "note": "a 41 MB launch is latency-bound (launch + 64-round chain + the longest unit program), "
                               "not bandwidth-bound; the fraction is reported because north_star states its target on this workload"}
    res["whole_region_verify"] = {
        "violations": vrep["violations"], "checks": vrep["checks"], "kernel_ms": vrep["kernel_ms"],
        "note": "hsw_gadget_verify: blocks + frames + links of the whole region, column image, on the device"}
    res["whole_region_to_host"] = {
        "ms_per_synthesis": float(np.median(tdl)) * 1e3, "bytes": (int(vw.gate_cells) + int(vw.lookup_cells)) * 32,
        "note": "synthesis + D2H of the used rows of the 9 gate columns and the lookup column into pinned memory (PCIe-bound); the chip columns would add 2 x 2 x 32,960 cells"}
    if tdc:
        res["whole_region_to_host_compact"] = {
            "ms_per_synthesis": float(np.median(tdc)) * 1e3, "bytes": (int(vw.gate_cells) + int(vw.lookup_cells)) * 8 + 48 * n_wide,
            "wide_cells": n_wide,
            "note": "hsw_gadget_download_region_compact: the same region as 8-byte cells + a side list of the cells wider than 64 bits; widened on the host (hsw_region_widen) it equals the 32-byte image bit for bit (checked here)"}
    else:
        res["whole_region_to_host_compact"] = {"error": n_wide}
    cfgw.close()
    # the same region in Montgomery form -- what bn256::Fr holds in memory, so what the Rust shim
    # (rust/reference-patch/src/hsw.rs) hands to region.assign_advice without a conversion per cell
    try:
        cfgm = hsw.Sha256DynamicConfig(eng_i, [1024], True, whole_digest=True)
        cfgm.set_repr(hsw._native.HSW_REPR_MONTGOMERY)
        cfgm.set_columns((1 << 17) - 9)
        for _ in range(6):
            assert L.hsw_gadget_reset(cfgm.h) == 0 and L.hsw_gadget_digest(cfgm.h, mbuf, 56, 0, C.byref(hres)) == 0
        tm = []
        for _ in range(200):
            assert L.hsw_gadget_reset(cfgm.h) == 0
            t1 = time.perf_counter()
            rcm = L.hsw_gadget_digest(cfgm.h, mbuf, 56, 0, C.byref(hres))
            tm.append(time.perf_counter() - t1)
            assert rcm == 0
        assert bytes(hres.output_bytes) == hashlib.sha256(m56).digest()
        lm = eng_i.last_launch()
        vm = cfgm.verify()
        res["whole_region_montgomery"] = {
            "ms_per_synthesis": float(np.median(tm)) * 1e3, "ms_min": float(np.min(tm)) * 1e3,
            "ms_p90": float(np.percentile(tm, 90)) * 1e3, "blocks_per_s": 16 / float(np.median(tm)),
            "GBps": region_bytes / float(np.median(tm)) / 1e9, "frac_of_peak": region_bytes / float(np.median(tm)) / 1e9 / HBM_PEAK_GBS,
            "kernel": lm["kernel"], "waves_per_block": lm["parts"], "grid": lm["grid"],
            "verify_on_device": {"violations": vm["violations"], "checks": vm["checks"]},
            "note": "the same synthesis with cells in Montgomery form (x * 2^256 mod p, halo2curves' in-memory Fr): "
                    "the form the Rust shim consumes"}
        # ... and on to the host as DISTINCT values: only the new witnesses (~40 % of the cells) cross PCIe, the
        # input-independent tape rebuilds copies and constants on the host (hsw_replay.cpp)
        try:
            tape = hsw._native.RegionTape()
            t1 = time.perf_counter()
            assert L.hsw_gadget_region_tape(cfgm.h, C.byref(tape)) == 0      # first call: builds the tape (once per circuit)
            tape_build_ms = (time.perf_counter() - t1) * 1e3
            himg = cfgm.download_region(pinned=True)                 # the 32-byte delivery, as the reference image
            dbuf = eng_i.host_empty((int(tape.distinct_capacity), 4))
            rgate, rlook = eng_i.host_empty(himg["gate"].shape), eng_i.host_empty(himg["lookup"].shape)
            rgate[:] = 0
            rdst = hsw._native.RegionHost(rgate.ctypes.data, rlook.ctypes.data, None, None)
            nd = C.c_size_t()
            threads = min(16, host_cores())
            t_full, t_dist, t_rep = [], [], []
            fdst = hsw._native.RegionHost(himg["gate"].ctypes.data, himg["lookup"].ctypes.data, None, None)
            for _ in range(9):
                assert L.hsw_gadget_reset(cfgm.h) == 0
                t1 = time.perf_counter()
                assert L.hsw_gadget_digest(cfgm.h, mbuf, 56, 0, C.byref(hres)) == 0
                assert L.hsw_gadget_download_region(cfgm.h, C.byref(fdst)) == 0
                t_full.append(time.perf_counter() - t1)
                assert L.hsw_gadget_reset(cfgm.h) == 0
                t1 = time.perf_counter()
                assert L.hsw_gadget_digest(cfgm.h, mbuf, 56, 0, C.byref(hres)) == 0
                assert L.hsw_gadget_download_region_distinct(cfgm.h, dbuf.ctypes.data, dbuf.shape[0], C.byref(nd)) == 0
                t2 = time.perf_counter()
                assert L.hsw_gadget_replay_region(cfgm.h, dbuf.ctypes.data, C.byref(rdst), threads) == 0
                t3 = time.perf_counter()
                t_dist.append(t2 - t1)
                t_rep.append(t3 - t2)
            assert np.array_equal(rgate, himg["gate"]) and np.array_equal(rlook, himg["lookup"]), "replayed region differs from the 32-byte delivery"
            res["whole_region_montgomery_to_host"] = {
                "full_32_byte_cells_ms": float(np.median(t_full)) * 1e3, "full_bytes": (int(vw.gate_cells) + int(vw.lookup_cells)) * 32,
                "distinct_ms": float(np.median(t_dist)) * 1e3, "distinct_cells": int(nd.value), "distinct_bytes": int(nd.value) * 32,
                "replay_ms": float(np.median(t_rep)) * 1e3, "replay_threads": threads,
                "tape_build_ms_once_per_circuit": tape_build_ms,
                "distinct_plus_replay_ms": float(np.median(np.array(t_dist) + np.array(t_rep))) * 1e3,
                "note": "synthesis + hsw_gadget_download_region_distinct (the new witnesses only, Montgomery form, pinned memory); "
                        "hsw_gadget_replay_region rebuilds gate image + lookup column on the host (checked bit-equal to the "
                        "32-byte delivery here); a consumer that walks the region cell by cell reads value(code[i]) through "
                        "the tape instead and needs no replay"}
        except Exception as ex:
            res["whole_region_montgomery_to_host"] = {"error": repr(ex)}
        cfgm.close()
    except Exception as ex:
        res["whole_region_montgomery"] = {"error": repr(ex)}
    # K proofs of this circuit in flight: K independent syntheses (each its own Context / region, frames included,
    # Montgomery cells) expanded by ONE hsw_gadget_digest_batch call -- north_star states its >= 60 % HBM target on
    # this workload, and one synthesis alone is a 41 MB, latency-bound launch
    batched = {}
    for K, form in ((8, "montgomery"), (64, "montgomery"), (256, "montgomery"), (512, "montgomery"), (256, "canonical")):
        try:
            bufs = [(C.c_uint8 * 56).from_buffer_copy(m56) for _ in range(K)]
            ptrs = (C.c_void_p * K)(*[C.addressof(b) for b in bufs])
            lens_ = (C.c_size_t * K)(*([56] * K))
            pres_ = (C.c_size_t * K)(*([0] * K))
            resv = (hsw._native.HashResult * K)()
            cfgk = hsw.Sha256DynamicConfig(eng_i, [1024] * K, True, whole_digest=True, independent=True)
            if form == "montgomery":
                cfgk.set_repr(hsw._native.HSW_REPR_MONTGOMERY)
            # placement (see roofline.placement): hsw_gadget_place tries three allocations of the chip columns for
            # the HBM-bound batches and keeps the one the gadget's own batch runs fastest on
            placed_ms = cfgk.place(3)[0] if K >= 64 else None
            tk = []
            for i in range(3 + (9 if K <= 64 else 5)):
                assert L.hsw_gadget_reset(cfgk.h) == 0
                t1 = time.perf_counter()
                rck = L.hsw_gadget_digest_batch(cfgk.h, K, ptrs, lens_, pres_, resv)
                tk.append(time.perf_counter() - t1)
                assert rck == 0
            dtk = float(np.median(tk[3:]))
            assert bytes(resv[K - 1].output_bytes) == hashlib.sha256(m56).digest()
            lk = eng_i.last_launch()
            vk = cfgk.verify()
            batched[str(K) if form == "montgomery" else "%d_%s" % (K, form)] = {
                "syntheses": K, "cells": form, "blocks": 16 * K, "ms": dtk * 1e3, "ms_per_synthesis": dtk * 1e3 / K, "blocks_per_s": 16 * K / dtk,
                "GBps": K * region_bytes / dtk / 1e9, "frac_of_peak": K * region_bytes / dtk / 1e9 / HBM_PEAK_GBS,
                "kernel": lk["kernel"], "grid": lk["grid"], "verify_on_device": {"violations": vk["violations"], "checks": vk["checks"]},
                "placement_candidates_batch_ms": placed_ms}
            cfgk.close()
        except Exception as ex:
            batched[str(K) if form == "montgomery" else "%d_%s" % (K, form)] = {"error": repr(ex)}
    res["batched"] = dict(batched, note="K independent syntheses of the bench circuit (HSW_GADGET_INDEPENDENT: a Context, zero cell, "
                          "lookup and chip rows of its own each; linear region streams) through one hsw_gadget_digest_batch call, "
                          "Montgomery cells, host padding + chain and the states back on the host included")
    eng_i.close()
    if with_cpu:
        try:
            cb = cpu_baseline_digest(m56, 1024, True)
            res["cpu_baseline"] = cb
            res["speedup_vs_cpu_1thread"] = cb["ms_per_synthesis"] / (dtw * 1e3)
            res["speedup_vs_cpu_all_cores"] = cb["all_cores"]["ms_per_synthesis_amortised"] / (dtw * 1e3)
            for K, b in batched.items():
                if isinstance(b, dict) and "ms" in b:
                    b["cpu_1thread_ms_for_K"] = cb["ms_per_synthesis"] * b["syntheses"]
                    b["speedup_vs_cpu_1thread"] = cb["ms_per_synthesis"] * b["syntheses"] / b["ms"]
        except Exception as ex:
            res["cpu_baseline"] = {"error": repr(ex)}
    return res


def extra_config2_regions(hsw, local_rank, msgs, n):
    """configs[2] as whole regions: 4,096 single-block digests, each with its prologue / epilogue cells
    (551 per digest) and the lookup column -- ONE framed expansion launch + one frame launch."""
    import ctypes as C
    import hashlib
    eng_i = hsw.WitnessEngine(local_rank, 8, 2, mode=hsw._native.HSW_MODE_HALO2_INTERNALS)
    nd = n
    bufs = [(C.c_uint8 * len(mm)).from_buffer_copy(mm.tobytes()) for mm in msgs[:nd]]
    ptrs = (C.c_void_p * nd)(*[C.addressof(b) for b in bufs])
    lens_ = (C.c_size_t * nd)(*[len(mm) for mm in msgs[:nd]])
    pres_ = (C.c_size_t * nd)(*([0] * nd))
    resv = (hsw._native.HashResult * nd)()
    cfgb = hsw.Sha256DynamicConfig(eng_i, [64] * nd, False, whole_digest=True)
    tsb = []
    for i in range(6):
        cfgb.reset()
        t1 = time.perf_counter()
        rcb = eng_i.lib.hsw_gadget_digest_batch(cfgb.h, nd, ptrs, lens_, pres_, resv)
        tsb.append(time.perf_counter() - t1)
        assert rcb == 0
    dtb = float(np.median(tsb[2:]))
    assert bytes(resv[5].output_bytes) == hashlib.sha256(msgs[5].tobytes()).digest()
    vb = cfgb.view()
    out = {"digests": nd, "ms": dtb * 1e3, "digests_per_s": nd / dtb, "gate_cells": int(vb.gate_cells),
           "lookup_cells": int(vb.lookup_cells),
           "note": "host padding + H2D + chain + framed expansion + frames + D2H of the states, through hsw_gadget_digest_batch"}
    cfgb.close()
    eng_i.close()
    return out


def extra_config4_substitute(hsw, eng, local_rank, blocks_h, pre_h, alg_bytes):
    """BASELINE configs[4] needs the Rust prover (create_proof at k=20): not runnable here.  SURVEY 8d
    substitute: the witness columns of a k=20-sized circuit (~120 blocks at 9 advice columns) delivered to
    HOST memory, where a CPU MSM/FFT prover would read them.  PCIe-bound by construction."""
    import ctypes as C
    import hashlib
    res = {}
    nb = 120
    t1 = time.perf_counter()
    hostout = eng.witness_blocks_host(blocks_h[:nb], pre_h[:nb], cursor0=0, pinned=True)
    t_first = time.perf_counter() - t1
    keep = hostout          # reuse the pinned buffers: time steady-state calls through the C ABI
    reps = 5
    t1 = time.perf_counter()
    for _ in range(reps):
        rc = eng.lib.hsw_witness_blocks_host(
            eng.h, blocks_h[:nb].ctypes.data, pre_h[:nb].ctypes.data, nb, 0, keep["gate"].ctypes.data,
            keep["dense"].ctypes.data, keep["spread"].ctypes.data, keep["dense"].shape[1], None, 0)
        assert rc == 0
    dt = (time.perf_counter() - t1) / reps
    res["block_streams_to_host"] = {
        "blocks": nb, "ms": dt * 1e3, "blocks_per_s": nb / dt, "host_GBps": nb * alg_bytes / dt / 1e9,
        "first_call_ms": t_first * 1e3,
        "note": "create_proof itself is not runnable (no Rust toolchain); pinned host buffers, PCIe Gen5 x16 spec 63 GB/s"}
    del hostout, keep
    # same delivery in the 8-byte transport form (HSW_REPR_COMPACT64): 4x fewer bytes over PCIe
    hc = eng.witness_blocks_host(blocks_h[:nb], pre_h[:nb], cursor0=0, pinned=True, flags=hsw.HSW_REPR_COMPACT64)
    t1 = time.perf_counter()
    for _ in range(reps):
        rc = eng.lib.hsw_witness_blocks_host(
            eng.h, blocks_h[:nb].ctypes.data, pre_h[:nb].ctypes.data, nb, 0, hc["gate"].ctypes.data,
            hc["dense"].ctypes.data, hc["spread"].ctypes.data, hc["dense"].shape[1], None, hsw.HSW_REPR_COMPACT64)
        assert rc == 0
    dtc = (time.perf_counter() - t1) / reps
    res["block_streams_to_host"]["compact64_transport"] = {
        "ms": dtc * 1e3, "blocks_per_s": nb / dtc, "host_GBps": nb * (alg_bytes // 4) / dtc / 1e9}
    del hc
    # the same circuit size as ONE whole region: a k = 20 context (max_rows = 2^20 - 9, lib.rs:351-360) of
    # eight digests summing to 120 blocks, input range checks on -- the column image of every advice cell
    try:
        eng_i = hsw.WitnessEngine(local_rank, 8, 2, mode=hsw._native.HSW_MODE_HALO2_INTERNALS)
        sizes = [1024] * 7 + [512]
        msgs = [bytes(((i * 7 + 3 * k) % 256) for i in range(s - 9 - 5 * k)) for k, s in enumerate(sizes)]
        cfg = hsw.Sha256DynamicConfig(eng_i, sizes, True, whole_digest=True)
        ncol = cfg.set_columns((1 << 20) - 9)
        for _ in range(2):
            cfg.reset()
            rs = cfg.digest_batch(msgs)
        ts = []
        for _ in range(10):
            cfg.reset()
            t1 = time.perf_counter()
            rs = cfg.digest_batch(msgs)
            ts.append(time.perf_counter() - t1)
        assert all(r.output_bytes == hashlib.sha256(m).digest() for r, m in zip(rs, msgs))
        # the same call through the C ABI on prebuilt arguments (the Python wrapper's marshalling of eight
        # messages is plumbing, not the product)
        Nn = hsw._native
        bufs = [(C.c_uint8 * len(m)).from_buffer_copy(m) for m in msgs]
        ptrs = (C.c_void_p * len(msgs))(*[C.addressof(b) for b in bufs])
        lens = (C.c_size_t * len(msgs))(*[len(m) for m in msgs])
        pres = (C.c_size_t * len(msgs))(*([0] * len(msgs)))
        hres = (Nn.HashResult * len(msgs))()
        tc = []
        for _ in range(30):
            cfg.reset()
            t1 = time.perf_counter()
            rcb = eng_i.lib.hsw_gadget_digest_batch(cfg.h, len(msgs), ptrs, lens, pres, hres)
            tc.append(time.perf_counter() - t1)
            assert rcb == 0
        assert bytes(hres[0].output_bytes) == hashlib.sha256(msgs[0]).digest()
        cfg.reset()
        rs = cfg.digest_batch(msgs)
        v = cfg.view()
        img = cfg.download_region(pinned=True)
        dst = hsw._native.RegionHost(img["gate"].ctypes.data, img["lookup"].ctypes.data,
                                     img["dense"].ctypes.data, img["spread"].ctypes.data)
        td = []
        for _ in range(5):
            cfg.reset()
            t1 = time.perf_counter()
            cfg.digest_batch(msgs)
            assert eng_i.lib.hsw_gadget_download_region(cfg.h, C.byref(dst)) == 0
            td.append(time.perf_counter() - t1)
        # the same region as distinct values in Montgomery form (what a halo2 prover takes): new witnesses only
        distinct = None
        try:
            cfg.reset()
            cfg.set_repr(hsw._native.HSW_REPR_MONTGOMERY)
            cfg.digest_batch(msgs)
            tape = hsw._native.RegionTape()
            assert eng_i.lib.hsw_gadget_region_tape(cfg.h, C.byref(tape)) == 0
            dbuf = eng_i.host_empty((int(tape.distinct_capacity), 4))
            nd = C.c_size_t()
            tdd, tdf = [], []
            for _ in range(5):
                cfg.reset()
                t1 = time.perf_counter()
                cfg.digest_batch(msgs)
                assert eng_i.lib.hsw_gadget_download_region_distinct(cfg.h, dbuf.ctypes.data, dbuf.shape[0], C.byref(nd)) == 0
                tdd.append(time.perf_counter() - t1)
                cfg.reset()
                t1 = time.perf_counter()
                cfg.digest_batch(msgs)
                assert eng_i.lib.hsw_gadget_download_region(cfg.h, C.byref(dst)) == 0
                tdf.append(time.perf_counter() - t1)
            distinct = {"montgomery_full_ms": float(np.median(tdf)) * 1e3, "montgomery_distinct_ms": float(np.median(tdd)) * 1e3,
                        "distinct_cells": int(nd.value), "distinct_bytes": int(nd.value) * 32}
            cfg.reset()
            cfg.set_repr(hsw._native.HSW_REPR_CANONICAL)
            cfg.digest_batch(msgs)
        except Exception as ex:
            distinct = {"error": repr(ex)}
        vrep = cfg.verify()
        nbytes = (int(v.gate_cells) + int(v.lookup_cells) + 2 * int(v.num_limb_sum)) * 32
        res["whole_region_k20"] = {
            "to_host_as_distinct_values": distinct,
            "digests": len(sizes), "blocks": sum(sizes) // 64, "advice_columns": ncol, "max_rows": (1 << 20) - 9,
            "gate_cells": int(v.gate_cells), "lookup_cells": int(v.lookup_cells),
            "synthesis_ms": float(np.median(ts)) * 1e3, "synthesis_GBps": nbytes / float(np.median(ts)) / 1e9,
            "synthesis_ms_c_abi": float(np.median(tc)) * 1e3, "synthesis_c_abi_GBps": nbytes / float(np.median(tc)) / 1e9,
            "synthesis_plus_download_ms": float(np.median(td)) * 1e3,
            "host_GBps": nbytes / float(np.median(td)) / 1e9,
            "verify": {"violations": vrep["violations"], "checks": vrep["checks"], "kernel_ms": vrep["kernel_ms"]},
            "note": "the advice-column image of a k = 20 region (8 digests, 120 blocks, input range checks) synthesised in HBM, "
                    "then gate + lookup + chip columns delivered to pinned host memory for a CPU prover"}
        cfg.close()
        eng_i.close()
    except Exception as ex:
        res["whole_region_k20"] = {"error": repr(ex)}
    return res


def extra_multi_gpu(dist, sh, eng, blocks, pre, out, n, world, rank, value):
    """north_star's all-gather of witness columns over xGMI (RCCL), on a bounded shard: 256 blocks (543 MB of
    gate cells) per rank.  Reported separately, never in `value`.  Symmetric on all ranks; a failure is
    recorded, not fatal."""
    import torch
    res = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
           "kernel_only_blocks_per_s": value}
    try:
        nb = min(256, n)
        oshard = eng.alloc_outputs(nb, 0)
        eng.witness_blocks(blocks[:nb], pre[:nb], cursor0=0, out=oshard)
        shard = oshard["gate"]
        counts = [nb] * world
        gathered = sh.allgather_gate(dist, shard, counts, eng.G)
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            eng.witness_blocks(blocks[:nb], pre[:nb], cursor0=0, out=oshard)    # the shard's kernel ...
            gathered = sh.allgather_gate(dist, shard, counts, eng.G)            # ... + its all-gather
        torch.cuda.synchronize()
        dist.barrier()
        dt = (time.perf_counter() - t1) / reps
        shard_bytes = shard.numel() * 8
        same = bool((gathered[rank * nb * eng.G:(rank + 1) * nb * eng.G] == shard).all())
        res["allgather"] = {"shard_blocks": nb, "shard_bytes": shard_bytes, "ms": dt * 1e3, "own_shard_intact": same,
                            "recv_GBps_per_gpu": shard_bytes * (world - 1) / dt / 1e9}
        res["kernel_plus_allgather_blocks_per_s"] = nb * world / dt
        del gathered
    except Exception as ex:
        res["allgather"] = {"error": repr(ex)}
    # The cheaper equivalent: exchange the 96-byte seeds and let every GPU
    # re-expand all ranks' blocks into its own HBM.  Bounded: 1,024 blocks per rank.
    try:
        nb = min(1024, n)
        counts = [nb] * world
        allout = eng.alloc_outputs(nb * world, 0)

        def replicate():
            sb, sp = sh.allgather_seeds(dist, blocks[:nb], pre[:nb], counts)
            eng.witness_blocks(sb, sp, cursor0=0, out=allout)
        replicate()
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            replicate()
        torch.cuda.synchronize()
        dist.barrier()
        dt = (time.perf_counter() - t1) / reps
        res["replicate_by_seed_exchange"] = {
            "blocks_per_rank": nb, "ms": dt * 1e3, "blocks_per_s_replicated_on_every_gpu": nb * world / dt,
            "xgmi_bytes_per_block": 96, "note": "every GPU ends with all ranks' witness columns in its own HBM"}
        del allout
    except Exception as ex:
        res["replicate_by_seed_exchange"] = {"error": repr(ex)}
    return res


# ------------------------------------------------------------------ one rank
def run_rank(args):
    # ONE JSON line on stdout: libraries that print there on their own (RCCL's version banner) go to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and "WORLD_SIZE" in os.environ and args.gpus != 1:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    distributed = world > 1 or os.environ.get("HSW_BENCH_FORCE_DIST") == "1"   # force: 1-rank RCCL rehearsal
    # Rehearsal knobs (not used by the driver): HSW_BENCH_BACKEND=gloo and HSW_BENCH_SAME_DEVICE=1 run
    # the multi-rank control flow on a box with ONE GPU (all ranks on cuda:0, collectives on the CPU).
    backend = os.environ.get("HSW_BENCH_BACKEND", "nccl")
    if os.environ.get("HSW_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))               # (set by the launcher; defaults for the 1-rank rehearsal)
        os.environ.setdefault("WORLD_SIZE", str(world))
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL over xGMI
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    hsw = importlib.import_module("halo2-dynamic-sha256_amd")
    eng = hsw.WitnessEngine(local_rank, 8, 2)
    shape = eng.shape
    alg_bytes = int(shape.algorithmic_bytes_per_block)

    # ---- synthetic workload: 55-byte messages, seed 0xC3 (+rank) ----
    n = args.messages_per_gpu
    rng = np.random.default_rng(0xC3 + rank)
    msgs = rng.integers(0, 256, (n, 55), dtype=np.uint8)
    blocks_h = sha_pad_single_block(msgs)
    pre_h = np.tile(IV, (n, 1))
    blocks = torch.from_numpy(blocks_h).to(dev)
    pre = torch.from_numpy(pre_h.view(np.int32)).to(dev)
    sh = importlib.import_module("halo2-dynamic-sha256_amd.sharding")
    start, count = sh.shard_range(n * world, world, rank)          # contiguous shard of the global batch
    assert count == n
    cursor0 = sh.shard_cursor(0, start, eng.limb_calls)            # rows land where the serial reference would put them
    # Where the gate stream and the chip columns sit decides between 1.58 and 1.78 ms per launch on this part
    # (profiles/r03_placement_probe.log; nothing in user space predicts it), so -- like a prover that allocates its
    # witness buffers once -- the launch is timed on a few candidate allocations (gate stream: hsw_device_alloc
    # ranges and plain buffers; chip columns: plain buffers and ranges of smaller pieces) and the best combination
    # is kept.  --placement-candidates 1 = take the first allocation as it comes (plain buffers).
    one = args.placement_candidates <= 1
    try:
        out, placement = eng.alloc_outputs_placed(blocks, pre, cursor0=cursor0, candidates=max(args.placement_candidates, 1),
                                                  gate_candidates=1 if one else 3, ranged=not one)
    except Exception as ex:                  # whatever goes wrong while choosing: plain buffers as they come
        out, placement = eng.alloc_outputs(n, cursor0), {"error": repr(ex)}
    eng.set_timing(True)

    def step():
        eng.witness_blocks(blocks, pre, cursor0=cursor0, out=out)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events on the stream the kernel is launched on (the engine was created on
    # torch's current stream, so torch's events are recorded on exactly that stream):
    # one pair per launch, read back after the timed region.
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    assert eng.stream.cuda_stream == torch.cuda.current_stream(dev).cuda_stream
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record(eng.stream)
        step()
        ev[i][1].record(eng.stream)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms_avg = float(np.mean([a.elapsed_time(b) for a, b in ev]))     # avg launch duration over the timed region
    launched = eng.last_launch()                                          # the instantiation that was timed
    # cross-check with the engine's own event pair (hsw_last_kernel_ms) on one more launch
    step()
    kernel_ms_engine = eng.last_kernel_ms()
    # practical write ceiling of this device/allocation: plain 16 B/lane fill of the same gate buffer
    fill_ms = min(eng.fill_calibrate(out["gate"]) for _ in range(3))
    fill_gbs = out["gate"].numel() * 8 / (fill_ms * 1e-3) / 1e9
    for _ in range(2):     # the fill clobbered the stream: regenerate before the checks below
        step()
    torch.cuda.synchronize()

    # spot-check (size-independent property): digest from next_states == hashlib
    import hashlib
    ns = out["next_states"][:4].cpu().numpy().view(np.uint32)
    for i in range(4):
        dig = b"".join(int(x).to_bytes(4, "big") for x in ns[i])
        assert dig == hashlib.sha256(msgs[i].tobytes()).digest(), "GPU digest mismatch"

    t_el = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
    # every rank's own kernel time (HIP events on its launch stream), gathered with one collective of a float each:
    # an 8-GPU line then shows a straggler instead of hiding it in the max-over-ranks step time
    kms = torch.tensor([kernel_ms_avg], dtype=torch.float64, device=coll_dev)
    kms_all = [kms]
    if distributed:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        kms_all = [torch.zeros_like(kms) for _ in range(world)]
        dist.all_gather(kms_all, kms)
    kernel_ms_per_rank = [float(t.item()) for t in kms_all]
    elapsed = float(t_el.item())
    total_blocks = n * world * args.steps
    value = total_blocks / elapsed

    extra = {}
    if not args.no_extra and rank == 0:
        extra.update(extra_representations(hsw, eng, blocks, pre, cursor0, out, n, alg_bytes, step))
        try:
            extra["config1_1KiB_message_16_blocks"] = extra_config1(hsw, eng, dev, local_rank, alg_bytes)
        except Exception as ex:
            extra["config1_1KiB_message_16_blocks"] = {"error": repr(ex)}
        try:
            extra["config0_benches_digest_rs"] = extra_config0(hsw, local_rank, not args.no_cpu_baseline)
        except Exception as ex:
            extra["config0_benches_digest_rs"] = {"error": repr(ex)}
        try:
            extra["config2_as_whole_regions"] = extra_config2_regions(hsw, local_rank, msgs, n)
        except Exception as ex:
            extra["config2_as_whole_regions"] = {"error": repr(ex)}
        try:
            extra["config4_substitute_k20"] = extra_config4_substitute(hsw, eng, local_rank, blocks_h, pre_h, alg_bytes)
        except Exception as ex:
            extra["config4_substitute_k20"] = {"error": repr(ex)}
    if distributed and not args.no_extra:
        mg = extra_multi_gpu(dist, sh, eng, blocks, pre, out, n, world, rank, value)
        if rank == 0:
            extra["multi_gpu"] = mg
    rccl_ranks = dist.get_world_size() if distributed else 1
    if distributed:
        dist.barrier()
        dist.destroy_process_group()      # the other ranks are done; rank 0 goes on to the CPU baseline alone

    if rank == 0:
        achieved = alg_bytes * n / (kernel_ms_avg * 1e-3) / 1e9
        traffic, traffic_src = None, None
        try:   # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE)
            for name in ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
                path = os.path.join(ROOT, "profiles", name)
                if not os.path.exists(path):
                    continue
                pmc = json.load(open(path))
                if pmc["algorithmic_bytes_per_launch"] == alg_bytes * n:      # same workload as this run
                    traffic = pmc["hbm_traffic_bytes_per_launch"]
                    traffic_src = "profiles/%s (rocprofv3 --pmc, separate passes; bytes per launch; kernel %s, measured at commit %s)" % (
                        name, pmc.get("kernel", "?"), pmc.get("commit", "?"))
                break
        except Exception:
            pass
        result = {
            "metric": "SHA256 compression blocks/sec (witness assign), k=17 shape",
            "value": value,
            "unit": "blocks/s",
            "n_gpus": world,
            "rccl_ranks": rccl_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32/u64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: %d independent single-block (55-byte) messages per GPU, "
                            "seed 0xC3+rank, pre-state = IV; num_bits_lookup=8, num_advice_columns=2" % n,
                "blocks_per_gpu": n,
                "bytes_per_block": alg_bytes,
                "cell_format": "256-bit BN254 Fr, canonical little-endian limbs",
                "output_bytes_per_step_per_gpu": alg_bytes * n,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBS,
                "achievable": HBM_ACHIEVABLE_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "calibrated_fill_GBps": fill_gbs,
                "calibrated_fill_note": "hsw_fill_calibrate on the same gate buffer: every wave streams its own contiguous 64 KiB "
                                        "chunks, the best pure-write pattern found on this part (tools/fillbench)",
                "placement": dict(placement, note="rank 0's: the launch timed on every combination of the candidate gate buffers "
                                  "(rows of kernel_ms) and candidate chip-column allocations (columns), the fastest kept: on MI355X the "
                                  "same launch takes between 1.58 and 1.78 ms depending on where its buffers sit "
                                  "(profiles/r03_placement_probe.log, DESIGN.md 5.1); the 'plain' row, first column, is what torch's "
                                  "allocator gives as it comes"),
                "kernel_ms_per_rank": {"min": min(kernel_ms_per_rank), "max": max(kernel_ms_per_rank), "ranks": kernel_ms_per_rank},
                "frac_per_rank": {"min": alg_bytes * n / (max(kernel_ms_per_rank) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "max": alg_bytes * n / (min(kernel_ms_per_rank) * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "kernel": "%s, %d waves per block%s" % (launched["kernel"], launched["parts"],
                                                         ", split phases" if launched["split"] else ""),
                "launch": launched,
                "kernel_ms": kernel_ms_avg,
                "kernel_ms_engine_events": kernel_ms_engine,
                "algorithmic_bytes_per_launch": alg_bytes * n,
            },
            "extra": extra,
        }
        mont = extra.get("montgomery_repr", {})
        if "kernel_ms" in mont:      # halo2curves' in-memory Fr: the form a Rust shim copies straight into the Region
            result["roofline"]["montgomery"] = {"kernel": mont["kernel"], "kernel_ms": mont["kernel_ms"],
                                                "achieved": mont["GBps"], "frac": mont["frac_of_peak"]}
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(blocks_h, pre_h)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--messages-per-gpu", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--placement-candidates", type=int, default=6,
                    help="candidate allocations of the chip columns (x 3 candidate gate buffers) timed before the run; 1 = plain buffers as they come")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))      # before anything touches HIP in this process
    run_rank(args)


if __name__ == "__main__":
    main()
