#!/usr/bin/env python3
"""Where prepare_full() (the tables of the nine-scalar job: Grams, edge cosines, node sets, 20 000 regressions) spends its host
time on a cold shard (dev tool)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wdg_amd import sweep, synth
jobs = sweep.make_jobs(synth.H_LEVELS_10_K10, range(5), k=10)
for rep in range(3):
    sb = sweep.SweepBatch(jobs, n_feat=500, gcn_hidden=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sb.prepare_full(epochs=100, sample_max=500, base_seed=rep)
    torch.cuda.synchronize(); print(f"prepare_full {1e3 * (time.perf_counter() - t0):.2f} ms")
sb = sweep.SweepBatch(jobs, n_feat=500, gcn_hidden=0)
pr = cProfile.Profile(); pr.enable()
sb.prepare_full(epochs=100, sample_max=500, base_seed=7)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
