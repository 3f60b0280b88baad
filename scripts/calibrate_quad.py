#!/usr/bin/env python3
"""Cost of a 16-row slice of the quad-row SpMM as a function of its width (entries per row): homogeneous batches (every graph
the same homophily level, i.e. the same width), one launch each -> us per launch and ns per (slice, feature group).
Feeds the cost model of ops._quad_unit_cost (the tape is cut into segments of equal modelled cost)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from wdg_amd import sweep, synth

rows = []
for k, levels in ((2, synth.H_LEVELS_10), (10, synth.H_LEVELS_10_K10)):
    for h in levels:
        jobs = sweep.make_jobs([h] * 10, range(8), k=k)  # 80 graphs of one width, ten per feature matrix
        batch = sweep.SweepBatch(jobs, n_feat=500, gcn_hidden=0)
        for _ in range(3):
            batch.spmm.launch()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            batch.spmm.launch()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        width = synth.out_degree(k, h) + 1
        slices = len(jobs) * 128 * 32
        rows.append((width, us, us * 1e3 * 256 * 16 / slices))
        print(f"k={k} h={h}: width {width:3d}  {us:7.1f} us per launch  {rows[-1][2]:7.1f} ns per slice and wave", flush=True)
        del batch
rows.sort()
print("width, ns per slice per wave:", [(w, round(c)) for w, _u, c in rows])
