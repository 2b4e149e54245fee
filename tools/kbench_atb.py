#!/usr/bin/env python
"""A^T B microbenchmark: split-K MFMA kernel vs rocBLAS through torch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kbench import timeit
from mdgrad_amd import ops
dev = "cuda:0"
print("%9s %4s %4s | %9s %9s %9s" % ("E", "M", "N", "hip us", "GB/s in", "torch us"))
for E, M, N in [(57344, 128, 30), (229376, 128, 30), (458752, 128, 30), (458752, 30, 30), (458752, 64, 128), (1835008, 128, 32)]:
    A = torch.randn(E, M, device=dev); B = torch.randn(E, N, device=dev)
    th = timeit(lambda: ops._atb(A, B), 20) * 1e3
    tt = timeit(lambda: A.t().matmul(B), 5) * 1e3
    print("%9d %4d %4d | %9.1f %9.1f %9.1f" % (E, M, N, th, 4.0 * E * (M + N) / th / 1e3, tt))
