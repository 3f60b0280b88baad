#!/usr/bin/env python3
"""The sweep's fused transform relu(Y W0) W1 alone (50 jobs, 2000 x 500 -> 64 -> 5): fp32 chain against split bf16 operands,
Y cold (flushed through a 1-GB fill) and warm (dev tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from wdg_amd import ops

n_jobs, m, k, h, c = 50, 2000, 500, 64, 5
pool = torch.randn(n_jobs, m, 512, device="cuda")
entries = []
for j in range(n_jobs):
    y = pool[j][:, :k]
    entries.append((y, torch.randn(k, h, device="cuda") * 0.05, None, torch.randn(h, c, device="cuda") * 0.1, None,
                    torch.zeros(m, 8, device="cuda")[:, :c]))
batch = ops.Mlp2Batch(entries, relu=True)
flush = torch.empty(1 << 28, device="cuda")


def timed(reps, cold):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reps)]
    for r in range(reps):
        if cold:
            flush.fill_(1.0)
        e[2 * r].record()
        batch.launch()
        e[2 * r + 1].record()
    torch.cuda.synchronize()
    t = sorted(e[2 * r].elapsed_time(e[2 * r + 1]) * 1e3 for r in range(reps))
    return t[len(t) // 2], t[0]


for parts in ("", "4", "5", "8"):
    for split in ("0", "1"):
        os.environ["WDG_MLP2_SPLIT"] = split
        if parts:
            os.environ["WDG_GEMM_PARTS"] = parts
        else:
            os.environ.pop("WDG_GEMM_PARTS", None)
        batch.launch()
        torch.cuda.synchronize()
        warm, cold = timed(20, False), timed(20, True)
        print(f"parts={parts or 'auto'} split={split}: warm median {warm[0]:.1f} us (min {warm[1]:.1f}), cold median {cold[0]:.1f} us (min {cold[1]:.1f})")
