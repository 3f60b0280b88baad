#!/usr/bin/env python3
"""Timing-only ablations of the row-lane SpMM on the bench workload (dev tool).  Outputs are WRONG under ablation.
reserved bit 0 = no Y stores, bit 1 = no X staging, bit 2 = no SELL sweep, bit 3 = register staging instead of LDS-DMA.
usage: ablate_rowlane.py [ablation codes ...]   (env WDG_SPMM_ORDER=0: job table in caller order)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from wdg_amd import ops, sweep, synth

jobs = sweep.make_jobs(synth.H_LEVELS_10, range(10), k=2)
names = {0: "full", 1: "no Y store", 2: "no X staging", 4: "no sweep", 3: "sweep only", 5: "staging only",
         6: "stores only", 7: "nothing", 8: "full, register staging"}
codes = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 3, 5, 6, 7, 8]
for ab in codes:
    ops.ABLATE_BITS = ab
    batch = sweep.SweepBatch(jobs, n_feat=500)
    for _ in range(3):
        batch.spmm.launch()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        batch.spmm.launch()
    b.record()
    torch.cuda.synchronize()
    print(f"ablate={ab} ({names.get(ab, '?'):22s}): {a.elapsed_time(b) / 20 * 1e3:8.1f} us  plan={batch.spmm.plan()}", flush=True)
    del batch
