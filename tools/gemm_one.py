#!/usr/bin/env python3
"""One GEMM micro-benchmark configuration (for rocprofv3 --pmc passes): kernel M N K iters."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suitesparse_amd import engine
k, M, N, K, it = (int(x) for x in sys.argv[1:6])
for kv in sys.argv[6:]:
    a, b = kv.split("=")
    engine.set_tuning(int(a), int(b))
print(f"kernel {k} M={M} N={N} K={K}: {engine.gemm_rate(k, M, N, K, it):.2f} TF/s")
