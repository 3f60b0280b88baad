#!/usr/bin/env python3
"""How the row-lane SpMM's time scales with the number of graphs, per ablation (dev tool): fixed cost vs per-item cost."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from wdg_amd import ops, sweep, synth

codes = [int(a) for a in sys.argv[1:]] or [7, 0]
for ab in codes:
    ops.ABLATE_BITS = ab
    for seeds in (1, 2, 5, 10, 20):
        batch = sweep.SweepBatch(sweep.make_jobs(synth.H_LEVELS_10, range(seeds), k=2), n_feat=500)
        for _ in range(3):
            batch.spmm.launch()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            batch.spmm.launch()
        b.record()
        torch.cuda.synchronize()
        print(f"ablate={ab} graphs={10 * seeds:4d}: {a.elapsed_time(b) / 20 * 1e3:8.1f} us  plan={batch.spmm.plan()}", flush=True)
        del batch
