#!/usr/bin/env python3
"""Latency of Sha256DynamicConfig.digest through the C ABI (tuning aid)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hsw = importlib.import_module("halo2-dynamic-sha256_amd")
eng = hsw.WitnessEngine(0, 8, 2)
for maxb, msg in [(128, b"abc"), (1024, b"\x01" * 56), (4096, b"\x02" * 4000)]:
    n = 200
    cfg = hsw.Sha256DynamicConfig(eng, [maxb] * (n + 5), True)
    for _ in range(5):
        cfg.digest(msg)
    t = time.perf_counter()
    for _ in range(n):
        cfg.lib.hsw_gadget_digest  # noqa
        cfg.digest(msg)
    dt = (time.perf_counter() - t) / n
    print("max %5d B (%2d blocks): %.1f us per digest, %.0f blocks/s" % (maxb, maxb // 64, dt * 1e6, maxb // 64 / dt))
    cfg.close()
