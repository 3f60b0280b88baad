#!/usr/bin/env python3
"""Times the batched aggregation of the bench workloads (dev tool): python scripts/dev/time_sweep.py [seeds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from wdg_amd import sweep, synth

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for k, levels in ((2, synth.H_LEVELS_10), (10, synth.H_LEVELS_10_K10)):
    batch = sweep.SweepBatch(sweep.make_jobs(levels, range(seeds), k=k), n_feat=500)
    for _ in range(5):
        batch.spmm.launch()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30):
        batch.spmm.launch()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 30 * 1e3
    alg = batch.spmm_algorithmic_bytes()
    print(f"k={k:2d} plan={batch.spmm.plan()} graphs={len(batch.jobs)} edges={batch.edges}: {us:7.1f} us -> "
          f"{alg / us / 1e3:7.1f} GB/s algorithmic ({alg / us / 1e3 / 80:.1f}% of 8 TB/s); "
          f"unique bytes {batch.spmm_unique_bytes() / us / 1e3:7.1f} GB/s; {batch.edges / us:.0f} M edges/s", flush=True)
    del batch
