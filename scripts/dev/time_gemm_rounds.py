#!/usr/bin/env python3
"""B-resident GEMM: time vs rows per job (dev tool) - how long one 16-tile round of a workgroup takes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from wdg_amd import ops

k, n = 500, 64
for jobs, m in ((100, 512), (100, 1024), (100, 2000), (100, 2048), (128, 1024), (256, 512), (256, 1024), (50, 2048)):
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
    print(f"jobs={jobs:4d} M={m:5d}: {us:8.1f} us  {2.0 * m * k * n * jobs / us / 1e6:7.1f} TFLOP/s", flush=True)
