#!/usr/bin/env python3
"""Time the full nine-scalar sweep job batch (C3 shard): step (aggregation + counters + LAS) + Gram / arc-cosine kernels +
edge cosines + every kernel regression of every epoch.  usage: time_full_metrics.py [seeds] [epochs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from wdg_amd import sweep, synth

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 100
jobs = sweep.make_jobs(synth.H_LEVELS_10_K10, range(seeds), k=10)
sb = sweep.SweepBatch(jobs, n_feat=500, gcn_hidden=0)
t0 = time.perf_counter()
sb.prepare_full(epochs=epochs, sample_max=500)
print(f"prepare_full (host RNG node sets, tables): {time.perf_counter() - t0:.2f} s; {sb.kr.n_jobs} regressions", flush=True)


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


sb.step()
print(f"step {timed(sb.step):8.3f} ms   gram+maps {timed(sb.gram.launch):8.3f} ms   edge cosines {timed(sb.ge.launch):8.3f} ms   "
      f"regressions {timed(sb.kr.launch):8.3f} ms", flush=True)
full = timed(lambda: (sb.step(), sb.launch_full()))
t = time.perf_counter()
rows = sb.full_metrics()
print(f"whole nine-scalar batch {full:.2f} ms = {len(jobs) / full * 1e3:.0f} graphs/s (+ host tail {1e3 * (time.perf_counter() - t):.1f} ms); "
      f"mean KR_L {rows[:, 7].mean():.4f} KR_NL {rows[:, 8].mean():.4f} ge {rows[:, 6].mean():.4f}")
