#!/usr/bin/env python3
"""How fast do the vendor BLAS libraries run the four ESM-1b GEMM shapes?  (Reference point for the hand-written kernel only.)"""
import torch
dev = torch.device("cuda", 0)
M = 66048
for name, N, K in (("qkv", 3840, 1280), ("out", 1280, 1280), ("fc1", 5120, 1280), ("fc2", 1280, 5120)):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(100):
        c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        c = a @ w.t()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 300
    print("%s M=%d N=%d K=%d: %.3f ms  %.0f TFLOP/s" % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
