#!/usr/bin/env python3
"""The eight XCD segments of the headline aggregation after SweepBatch.tune(): their spans inside the step (device clock, us) - what
the slowest segment costs over the mean is what a perfect balance could still gain.  usage: segment_spans.py [k] [seeds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from wdg_amd import sweep, synth

k = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
levels = synth.H_LEVELS_10 if k == 2 else synth.H_LEVELS_10_K10
jobs = sweep.make_jobs(levels, range(seeds), k=k, n_nodes=int(os.environ.get("N", 2000)))
sb = sweep.SweepBatch(jobs, n_feat=500, gcn_hidden=64, tune=True)
sp = sb.spmm
print("tuned:", None if sp.tuned is None else (round(sp.tuned[0] * 1e3, 1), None if sp.tuned[1] is None else np.round(sp.tuned[1], 3).tolist()))
clock = sp.new_clock()
spans, starts, ends = np.zeros(8), np.zeros(8), np.zeros(8)
reps = 20
for _ in range(3):
    sb.step()
for _ in range(reps):
    sp.launch(clock=clock)
    sb.step_rest()
    torch.cuda.synchronize()
    c = clock.cpu().numpy().reshape(-1, 2).astype(np.float64) / 100.0  # us
    t0 = c[:, 0].min()
    for x in range(8):
        starts[x] += (c[x::8, 0].min() - t0) / reps
        ends[x] += (c[x::8, 1].max() - t0) / reps
        spans[x] += (c[x::8, 1].max() - c[x::8, 0].min()) / reps
    wg = c[:, 1] - c[:, 0]
print("per XCD: first workgroup start", np.round(starts, 1).tolist())
print("per XCD: last workgroup end   ", np.round(ends, 1).tolist())
print("workgroup spans of the last launch: mean %.1f min %.1f max %.1f us; launch span %.1f" % (wg.mean(), wg.min(), wg.max(), c[:, 1].max() - t0))
items = sp.items.cpu().numpy() if hasattr(sp, "items") else None
print("segments", sp.n_segments, "items", sp.n_items)
