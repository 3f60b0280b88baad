#!/usr/bin/env python3
"""How the aggregation launch and the step settle after a sync (dev tool): per-step HIP-event times of the first steps of a
timed region, the way bench.py --steps 20 --warmup 5 runs them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wdg_amd import sweep, synth
jobs = sweep.make_jobs(synth.H_LEVELS_10_K10, range(5), k=10)
b = sweep.SweepBatch(jobs, n_feat=500, tune=True)
for _ in range(5):
    b.step()
for trial in range(3):
    torch.cuda.synchronize()
    time.sleep(0.002)
    n = 40
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    t0 = time.perf_counter()
    for s in range(n):
        ev[s][0].record(); b.spmm.launch(); ev[s][1].record(); b.step_rest(); ev[s][2].record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    sp = [a.elapsed_time(c) * 1e3 for a, c, _ in ev]
    st = [ev[s][0].elapsed_time(ev[s + 1][0]) * 1e3 for s in range(n - 1)]
    print("trial", trial, "wall us/step", el / n * 1e6)
    print("  spmm us:", " ".join(f"{v:.0f}" for v in sp[:24]))
    print("  step us:", " ".join(f"{v:.0f}" for v in st[:24]))
