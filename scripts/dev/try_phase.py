#!/usr/bin/env python3
"""quad-row launch time vs the per-phase cost assumed by the tape cut (dev tool)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wdg_amd import sweep, synth
for k, seeds in ((10, 5), (2, 10), (10, 10)):
    levels = synth.H_LEVELS_10 if k == 2 else synth.H_LEVELS_10_K10
    jobs = sweep.make_jobs(levels, range(seeds), k=k)
    for ph in (0, 6000, 9000, 12000, 15000, 18000):
        os.environ["WDG_QUAD_PHASE_NS"] = str(ph)
        b = sweep.SweepBatch(jobs, n_feat=500, gcn_hidden=0)
        for _ in range(5): b.spmm.launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): b.spmm.launch()
        e1.record(); torch.cuda.synchronize()
        print(f"k={k} seeds={seeds} phase_ns={ph:6d}: {e0.elapsed_time(e1) / 30 * 1e3:7.1f} us  items={getattr(b.spmm, 'n_items', None)}", flush=True)
