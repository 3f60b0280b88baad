#!/usr/bin/env python3
"""A/B on one box: quad-row launch time (HIP events, interleaved repeats) for the phase order and the phase cost of the cut"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from wdg_amd import sweep, synth
variants = [("order=0 phase=0", "0", "0"), ("order=1 phase=0", "1", "0"), ("order=1 phase=6000", "1", "6000"),
            ("order=1 phase=10000", "1", "10000"), ("order=1 phase=14000", "1", "14000")]
for k, seeds in ((10, 5), (10, 10), (2, 10)):
    levels = synth.H_LEVELS_10 if k == 2 else synth.H_LEVELS_10_K10
    jobs = sweep.make_jobs(levels, range(seeds), k=k)
    batches = []
    for name, order, ph in variants:
        os.environ["WDG_QUAD_PHASE_ORDER"], os.environ["WDG_QUAD_PHASE_NS"] = order, ph
        batches.append(sweep.SweepBatch(jobs, n_feat=500, gcn_hidden=0))
    times = np.zeros((len(variants), 6))
    for rep in range(6):
        for vi, b in enumerate(batches):
            for _ in range(3): b.spmm.launch()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): b.spmm.launch()
            e1.record(); torch.cuda.synchronize()
            times[vi, rep] = e0.elapsed_time(e1) / 20 * 1e3
    for (name, _o, _p), t in zip(variants, times):
        print(f"k={k} seeds={seeds} {name:22s}: median {np.median(t):7.1f} us  min {t.min():7.1f}  max {t.max():7.1f}", flush=True)
    del batches
