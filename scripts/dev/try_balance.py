#!/usr/bin/env python3
"""what SweepBatch.tune (feedback balancing of the XCDs' segments) finds, per configuration (dev tool)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from wdg_amd import sweep, synth
for k, seeds in ((10, 5), (2, 10), (10, 10)):
    levels = synth.H_LEVELS_10 if k == 2 else synth.H_LEVELS_10_K10
    os.environ["WDG_QUAD_TUNE"] = "0"
    b = sweep.SweepBatch(sweep.make_jobs(levels, range(seeds), k=k), n_feat=500)
    os.environ["WDG_QUAD_TUNE"] = "1"
    sp = b.spmm
    clock = sp.new_clock()
    def timed(n=20):
        for _ in range(3): b.step()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        spans = np.zeros(8)
        for a, c in ev:
            a.record(); sp.launch(clock=clock); c.record(); b.step_rest()
            torch.cuda.synchronize(); spans += sp.segment_spans(clock) / n
        return sorted(a.elapsed_time(c) for a, c in ev)[n // 2] * 1e3, spans
    t0, s0 = timed()
    print(f"k={k} seeds={seeds} modelled cut : launch {t0:6.1f} us  spans " + " ".join(f"{v:5.1f}" for v in s0), flush=True)
    best = b.tune(rounds=8)
    t1, s1 = timed()
    print(f"k={k} seeds={seeds} balanced cut : launch {t1:6.1f} us  spans " + " ".join(f"{v:5.1f}" for v in s1)
          + f"   (tune's best {best[0] * 1e3:6.1f} us, shares " + ("equal" if best[1] is None else " ".join(f"{v:.2f}" for v in best[1])) + ")", flush=True)
