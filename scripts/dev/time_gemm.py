#!/usr/bin/env python3
"""GEMM batch time vs number of jobs (dev tool): M=2000, K=500, N=64 per job, like the sweep's relu(Y W0)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from wdg_amd import ops

m, k, n = 2000, 500, 64
for jobs in (1, 4, 16, 32, 64, 100, 200):
    a = [torch.randn(m, k, device="cuda") for _ in range(jobs)]
    b = [torch.randn(k, n, device="cuda") for _ in range(jobs)]
    c = [torch.empty(m, n, device="cuda") for _ in range(jobs)]
    batch = ops.GemmBatch(list(zip(a, b, c, [None] * jobs)), relu=True)
    for _ in range(3):
        batch.launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        batch.launch()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"jobs={jobs:4d} workgroups={jobs * 16:5d}: {us:8.1f} us  {2.0 * m * k * n * jobs / us / 1e6:7.1f} TFLOP/s  A stream {4.0 * m * k * jobs / us / 1e6:6.2f} TB/s", flush=True)
    if os.environ.get("WDG_GEMM_CHECK"):
        os.environ["WDG_GEMM_TILE"] = "1"
        c2 = [torch.empty(m, n, device="cuda") for _ in range(jobs)]
        ops.GemmBatch(list(zip(a, b, c2, [None] * jobs)), relu=True).launch()
        torch.cuda.synchronize()
        del os.environ["WDG_GEMM_TILE"]
        print("   bitwise equal to the tile kernel:", all(torch.equal(x, y) for x, y in zip(c, c2)), flush=True)
