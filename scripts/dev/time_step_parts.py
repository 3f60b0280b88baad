#!/usr/bin/env python3
"""Times every launch of a sweep step on the bench workload (dev tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from wdg_amd import sweep, synth

k = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
batch = sweep.SweepBatch(sweep.make_jobs(synth.H_LEVELS_10 if k == 2 else synth.H_LEVELS_10_K10, range(seeds), k=k), n_feat=500)
parts = [(f"spmm (F={batch.agg_feat})", batch.spmm), ("edge stats", batch.stats), ("spmm label aggregation (F=C)", batch.spmm_las),
         ("las", batch.las), ("fused relu(Y W0) W1", batch.gcn["mlp"]), ("gemm1 relu(Y W0)", batch.gcn["gemm1"]),
         ("gemm2", batch.gcn["gemm2"]), ("spmm logits (F=C)", batch.gcn["spmm"])]
parts = [(n, p) for n, p in parts if p is not None]
for _ in range(3):
    batch.step()
torch.cuda.synchronize()
for name, part in parts:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        part.launch()
    b.record()
    torch.cuda.synchronize()
    plan = part.plan() if hasattr(part, "plan") else ""
    print(f"{name:32s} {a.elapsed_time(b) / 20 * 1e3:8.1f} us  {plan}", flush=True)
