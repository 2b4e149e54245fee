#!/usr/bin/env python
"""Microbenchmark of the MFMA continuous-filter kernel (K9, csrc/cfconv_filter.hip) and the graph
gather kernel (K10, cfconv_agg): time, f32-MFMA TFLOP/s, HBM GB/s (algorithmic bytes), against the
torch-op chain they replace.  Usage: python tools/kbench_filter.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kbench import timeit  # noqa: E402


def main():
    from mdgrad_amd import ops
    dev = "cuda:0"
    print("%9s %4s %4s | %9s %9s %9s %9s | %9s | %9s %9s" % ("E", "G", "F", "f32 us", "TFLOP/s", "GB/s out", "%HBM 8T", "torch us", "bf16 us", "GB/s out"))
    for E, G, F in [(57344, 30, 128), (57344, 32, 128), (57344, 64, 256), (458752, 30, 128), (1048576, 32, 128),
                    (1048576, 64, 256), (4194304, 32, 128)]:
        torch.manual_seed(0)
        d = torch.rand(E, device=dev) * 6.0
        mu = torch.linspace(0, 6.0, G, device=dev)
        width = torch.full((G,), 6.0 / (G - 1), device=dev)
        W1 = torch.randn(G, G, device=dev) / G ** 0.5
        b1 = torch.randn(G, device=dev) * 0.1
        W2 = torch.randn(F, G, device=dev) / G ** 0.5
        b2 = torch.randn(F, device=dev) * 0.1
        args = (d, mu, width, W1, b1, W2, b2)
        with torch.no_grad():
            t_hip = timeit(lambda: ops.CfconvFilterFn.apply(*args), 20) * 1e3
            t_ref = timeit(lambda: ops.filter_reference(*args), 10) * 1e3
            t_bf = timeit(lambda: ops.CfconvFilterFn.apply(*args, True), 20) * 1e3
            err = float((ops.CfconvFilterFn.apply(*args) - ops.filter_reference(*args)).abs().max())
            errb = float((ops.CfconvFilterFn.apply(*args, True) - ops.filter_reference(*args)).abs().max())
        flop = 2.0 * E * G * (G + F)
        byts = 4.0 * E * (F + 1)
        print("%9d %4d %4d | %9.1f %9.2f %9.1f %9.1f | %9.1f | %9.1f %9.1f   max|diff| f32 %.1e bf16 %.1e" % (
            E, G, F, t_hip, flop / t_hip / 1e6, byts / t_hip / 1e3, 100 * byts / t_hip / 1e3 / 8000.0, t_ref,
            t_bf, byts / t_bf / 1e3, err, errb))


if __name__ == "__main__":
    main()
