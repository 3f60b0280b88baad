#!/usr/bin/env python3
"""Where a cold shard's build time goes (dev tool): host concatenation, uploads, the batched build, tables."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from wdg_amd import ops, sweep, synth

jobs = sweep.make_jobs(synth.H_LEVELS_10_K10, range(5), k=10)
feats, inputs = {}, []
for j in jobs:
    src, dst, lab = synth.regular_graph(j.n_nodes, j.n_classes, j.k, j.h, j.seed)
    feats.setdefault(j.seed, synth.features(j.n_nodes, 500, j.seed))
    inputs.append((src, dst, lab, feats[j.seed]))
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    coos = [(i[0], i[1], 2000) for i in inputs]
    gb = ops.GraphBatch(coos, ops.COO_ADD_SELF_LOOPS, quad=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    d = gb.degree_norm(ops.NORM_RW, ops.PREC_F32)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    sb = sweep.SweepBatch(jobs, n_feat=500, gcn_hidden=0, inputs=inputs)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"GraphBatch {1e3 * (t1 - t0):.2f} ms, degree_norm {1e3 * (t2 - t1):.2f} ms, whole SweepBatch {1e3 * (t3 - t2):.2f} ms")
import cProfile
import pstats
pr = cProfile.Profile()
pr.enable()
sb = sweep.SweepBatch(jobs, n_feat=500, gcn_hidden=0, inputs=inputs)
torch.cuda.synchronize()
pr.disable()
def show(pr, key, n):
    st = pstats.Stats(pr)
    rows = sorted(st.stats.items(), key=lambda kv: -kv[1][3 if key == "cumulative" else 2])[:n]
    for (fn, line, name), (cc, nc, tt, ct, _) in rows:
        print(f"{nc:6d} calls  own {tt * 1e6:9.0f} us  cumulative {ct * 1e6:9.0f} us  {os.path.basename(fn)}:{line} {name}")


show(pr, "cumulative", 45)
pr = cProfile.Profile()
pr.enable()
coos = [(i[0], i[1], 2000) for i in inputs]
gb = ops.GraphBatch(coos, ops.COO_ADD_SELF_LOOPS, quad=True)
torch.cuda.synchronize()
pr.disable()
print()
show(pr, "tottime", 14)
