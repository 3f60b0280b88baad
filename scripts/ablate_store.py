#!/usr/bin/env python3
"""Store-granularity probe (dev tool): Y-store-only timing of the slab kernel at 32/64/128-byte row pieces (N=1000)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from wdg_amd import sweep, synth

jobs = sweep.make_jobs(synth.H_LEVELS_10, range(20), k=2, n_nodes=1000)
for slab, thr in ((8, 512), (16, 512), (16, 1024), (32, 1024)):
    os.environ["WDG_SPMM_SLAB"], os.environ["WDG_SPMM_THREADS"] = str(slab), str(thr)
    res = {}
    for ab in (6, 7, 0):
        os.environ["WDG_SPMM_ABLATE"] = str(ab)
        batch = sweep.SweepBatch(jobs, n_feat=512)
        for _ in range(3):
            batch.spmm.launch()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            batch.spmm.launch()
        b.record()
        torch.cuda.synchronize()
        res[ab] = a.elapsed_time(b) / 20 * 1e3
        ybytes = sum(y.numel() * 4 for y in batch.y)
        del batch
    print(f"slab={slab:2d} ({slab*4:3d} B pieces) thr={thr}: stores-only {res[6]:7.1f} us, floor {res[7]:6.1f} us -> "
          f"{ybytes / (res[6] - res[7]) / 1e6:7.1f} GB/s store rate; full kernel {res[0]:7.1f} us", flush=True)
