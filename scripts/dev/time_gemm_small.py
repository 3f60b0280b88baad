#!/usr/bin/env python3
"""Tile vs B-resident GEMM kernel over K for the sweep's job count (dev tool): 100 jobs, M = 2000."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from wdg_amd import ops

jobs, m = 100, 2000
for k, n in ((64, 5), (64, 64), (128, 64), (256, 64), (500, 5), (500, 64)):
    a = [torch.randn(m, k, device="cuda") for _ in range(jobs)]
    b = [torch.randn(k, n, device="cuda") for _ in range(jobs)]
    c = [torch.empty(m, n, device="cuda") for _ in range(jobs)]
    batch = ops.GemmBatch(list(zip(a, b, c, [None] * jobs)))
    res = []
    for env in ("WDG_GEMM_TILE", "WDG_GEMM_RESIDENT"):
        os.environ[env] = "1"
        for _ in range(3):
            batch.launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            batch.launch()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
        del os.environ[env]
    print(f"K={k:4d} N={n:3d}: tile {res[0]:7.1f} us   resident {res[1]:7.1f} us", flush=True)
