#!/usr/bin/env python3
"""Steady-state latency of Sha256DynamicConfig.digest through the C ABI: one context per circuit,
hsw_gadget_reset per synthesis pass (tuning aid)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hsw = importlib.import_module("halo2-dynamic-sha256_amd")
N = hsw._native
for mode, whole in [(0, False), (N.HSW_MODE_HALO2_INTERNALS, True)]:
    eng = hsw.WitnessEngine(0, 8, 2, mode=mode)
    for maxb, msg in [(128, b"abc"), (1024, b"\x01" * 56), (4096, b"\x02" * 4000), (16384, b"\x03" * 16000)]:
        cfg = hsw.Sha256DynamicConfig(eng, [maxb], True, whole_digest=whole)
        if whole:
            cfg.set_columns((1 << 17) - 9 if maxb <= 1024 else (1 << 21) - 9)
        for _ in range(5):
            cfg.reset(); cfg.digest(msg)
        n = 200
        t = time.perf_counter()
        for _ in range(n):
            cfg.reset(); cfg.digest(msg)
        dt = (time.perf_counter() - t) / n
        print("%-13s max %5d B (%3d blocks): %6.1f us per synthesis, %7.0f blocks/s" % (
            "whole region" if whole else "block streams", maxb, maxb // 64, dt * 1e6, maxb // 64 / dt), flush=True)
        cfg.close()
    eng.close()
