#!/usr/bin/env python3
"""Micro-benchmarks of the MFMA kernels on the GPU box (results go to profiles/)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import engine
print("mfma probe (TF/s, cycles per MFMA per wave @2 waves/SIMD, MHz):", engine.mfma_f64_probe())
for (M, N, K) in [(16384, 16384, 512), (16384, 16384, 128), (8192, 8192, 64), (8192, 8192, 256), (8192, 8192, 1024), (16384, 4096, 256), (16384, 256, 64),
                  (4096, 4096, 4096), (2048, 512, 128), (1024, 1024, 64), (512, 128, 32)]:
    r0 = engine.gemm_rate(0, M, N, K, 5)
    r1 = engine.gemm_rate(1, M, N, K, 5)
    print(f"M={M:6d} N={N:6d} K={K:5d}  k_gemm64 {r0:7.2f}  k_gemm128 {r1:7.2f} TF/s", flush=True)
