#!/usr/bin/env python3
"""Timeline of one synthesis of the reference's bench circuit (benches/digest.rs: 56-byte message, 16 blocks,
9 columns) through hsw_gadget_digest -- run under
    rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/tl -o tl -- python3 tools/region_timeline.py run
then summarise the rocpd database with
    python3 tools/region_timeline.py summarise gpurun_out/tl/.../tl_results.db
(per synthesis: start/end of every kernel and copy relative to the first event, gaps in between)."""
import importlib, os, sqlite3, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    hsw = importlib.import_module("halo2-dynamic-sha256_amd")
    N = hsw._native
    eng = hsw.WitnessEngine(0, 8, 2, mode=N.HSW_MODE_HALO2_INTERNALS)
    cfg = hsw.Sha256DynamicConfig(eng, [1024], True, whole_digest=True)
    cfg.set_columns((1 << 17) - 9)
    m = bytes([1] * 56)
    for _ in range(10):
        cfg.reset(); cfg.digest(m)
    ts = []
    for _ in range(40):
        cfg.reset()
        t = time.perf_counter()
        cfg.digest(m)
        ts.append(time.perf_counter() - t)
        time.sleep(0.002)            # a visible gap between syntheses in the trace
    ts.sort()
    print("whole region: median %.1f us, min %.1f us" % (ts[len(ts) // 2] * 1e6, ts[0] * 1e6))
    cfg.close()
    eng2 = hsw.WitnessEngine(0, 8, 2)
    cfg2 = hsw.Sha256DynamicConfig(eng2, [1024] * 64, True)
    for _ in range(10):
        cfg2.reset(); cfg2.digest(m)
    ts = []
    for _ in range(40):
        cfg2.reset()
        t = time.perf_counter()
        cfg2.digest(m)
        ts.append(time.perf_counter() - t)
        time.sleep(0.002)
    ts.sort()
    print("block streams: median %.1f us, min %.1f us" % (ts[len(ts) // 2] * 1e6, ts[0] * 1e6))


def run_k(k, mont):
    """K independent syntheses of the bench circuit in one hsw_gadget_digest_batch call (bench.py's `batched`)."""
    import numpy as np
    hsw = importlib.import_module("halo2-dynamic-sha256_amd")
    N = hsw._native
    eng = hsw.WitnessEngine(0, 8, 2, mode=N.HSW_MODE_HALO2_INTERNALS)
    cfg = hsw.Sha256DynamicConfig(eng, [1024] * k, True, whole_digest=True, independent=True)
    if mont:
        cfg.set_repr(N.HSW_REPR_MONTGOMERY)
    rng = np.random.default_rng(5)
    msgs = [rng.integers(0, 256, 56, dtype=np.uint8).tobytes() for _ in range(k)]
    for _ in range(3):
        cfg.reset(); cfg.digest_batch(msgs)
    ts = []
    for _ in range(8):
        cfg.reset()
        t = time.perf_counter()
        cfg.digest_batch(msgs)
        ts.append(time.perf_counter() - t)
        time.sleep(0.005)
    ts.sort()
    print("K = %d %s: median %.1f us, min %.1f us" % (k, "montgomery" if mont else "canonical", ts[len(ts) // 2] * 1e6, ts[0] * 1e6))
    cfg.close()


def summarise(db):
    cur = sqlite3.connect(db).cursor()
    ev = [(s, e, "K " + n.split("(")[0][-60:]) for n, s, e in cur.execute("select name, start, end from kernels")]
    try:
        ev += [(s, e, "C %s %d B" % (n, b)) for n, s, e, b in cur.execute("select name, start, end, size from memory_copies")]
    except Exception as ex:
        print("no memory_copies table:", ex)
    ev.sort()
    # group into syntheses: a gap > 1 ms starts a new group
    groups, cur_g = [], []
    for s, e, n in ev:
        if cur_g and s - cur_g[-1][1] > 1_000_000:
            groups.append(cur_g); cur_g = []
        cur_g.append((s, e, n))
    if cur_g:
        groups.append(cur_g)
    for g in (groups[-3:] if len(groups) < 60 else groups[-45:-38] + groups[-3:]):
        t0 = g[0][0]
        print("--- %d events, span %.1f us" % (len(g), (g[-1][1] - t0) / 1e3))
        for s, e, n in g:
            print("  +%7.1f us  %6.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    elif sys.argv[1] == "runk":
        run_k(int(sys.argv[2]), len(sys.argv) > 3 and sys.argv[3] == "montgomery")
    else:
        summarise(sys.argv[2])
