import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import numpy as np, torch
from wdg_amd import ops, sweep, synth
import cProfile, pstats
jobs = sweep.make_jobs(synth.H_LEVELS_10_K10, range(5), k=10)
coos = []
for j in jobs:
    src, dst, lab = synth.regular_graph(j.n_nodes, j.n_classes, j.k, j.h, j.seed)
    coos.append((src, dst, 2000))
for _ in range(3):
    gb = ops.GraphBatch(coos, ops.COO_ADD_SELF_LOOPS, quad=True, defer=True); torch.cuda.synchronize(); gb.finish()
gbs = [ops.GraphBatch(coos, ops.COO_ADD_SELF_LOOPS, quad=True, defer=True) for _ in range(5)]
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for gb in gbs: gb.finish()
dt = time.perf_counter() - t0
pr.disable()
print("finish ms", dt / 5 * 1e3)
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
