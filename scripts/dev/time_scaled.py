#!/usr/bin/env python3
"""Scaled single-graph aggregation (SURVEY 8(d) C3 'scaled variant'): N = 2^17 .. 2^20, k = 10, F = 128 / 512 - one SpMM
larger than L2 / Infinity Cache (dev tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from wdg_amd import ops

rng = np.random.default_rng(0)
for n, f in ((1 << 17, 128), (1 << 17, 512), (1 << 20, 128)):
    deg = 50  # int(10 / 0.2)
    src = np.repeat(np.arange(n, dtype=np.int64), deg)
    dst = rng.integers(0, n, n * deg)
    g = ops.CsrGraph.from_coo(src, dst, n, None, ops.COO_ADD_SELF_LOOPS | ops.COO_BINARISE)
    d = ops.degree_norm(g, ops.NORM_RW)["dinv"]
    x = torch.randn(n, f, device="cuda")
    y = ops.spmm(g, x, row_scale=d)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.spmm(g, x, row_scale=d, out=y)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    alg = 4 * (n + 1) + 4 * g.nnz + 4 * n + 8 * n * f
    print(f"N=2^{int(np.log2(n))} F={f} nnz={g.nnz}: {us:9.1f} us  {g.nnz / us / 1e3:6.2f} G edges/s  algorithmic {alg / us / 1e3:7.1f} GB/s "
          f"({alg / us / 1e3 / 80:.1f}% of 8 TB/s)  gathered {4.0 * g.nnz * f / us / 1e3:8.1f} GB/s  plan {ops.spmm_plan(n, n, f)}", flush=True)
