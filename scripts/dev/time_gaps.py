#!/usr/bin/env python3
"""Where the launch gaps of a sweep step come from (dev tool): step time with / without the per-launch timing events of
bench.py and with 1 / 2 / 3 streams."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from wdg_amd import sweep, synth

jobs = sweep.make_jobs(synth.H_LEVELS_10, range(10), k=2)
for streams in ("3", "2", "1"):
    os.environ["WDG_SWEEP_STREAMS"] = streams
    batch = sweep.SweepBatch(jobs, n_feat=500)
    for events in (True, False):
        for _ in range(10):
            batch.step()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
        t0 = time.perf_counter()
        for s in range(200):
            if events:
                ev[s][0].record()
            batch.spmm.launch()
            if events:
                ev[s][1].record()
            batch.step_rest()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 200 * 1e3
        print(f"streams={streams} timing events={events}: {dt:.4f} ms per step", flush=True)
    del batch
