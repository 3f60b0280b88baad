#!/usr/bin/env python3
"""Host time of a wide base's first batch (SweepBatch + prepare_full) against the number of adjacencies in the shard (dev tool): how
much of the start-up of a rank's share is fixed, how much per job."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from wdg_amd import ops, sweep, synth


class A:
    nodes, kr_epochs = 2000, 100


inp = bench.whole_inputs(A)
pairs, graphs, feats = inp["pairs"], inp["graphs"], inp["feats"]
name, fx, sample_max = feats[0]
width = next(iter(fx.values())).shape[1]
for n_jobs in (4, 8, 16, 35, 35, 8, 4):
    mine = pairs[:n_jobs]
    gi = [graphs[(j.h, j.seed)] for j in mine]
    res = []
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gb = ops.GraphBatch([(src, dst, j.n_nodes) for j, (src, dst, _l) in zip(mine, gi)], ops.COO_ADD_SELF_LOOPS, quad=True, defer=True)
        gb.finish()
        t1 = time.perf_counter()
        sb = sweep.SweepBatch(mine, n_feat=width, gcn_hidden=0, inputs=[(s, d, l, fx[j.seed]) for j, (s, d, l) in zip(mine, gi)],
                              labels_only=True, graph_batch=gb)
        t2 = time.perf_counter()
        sb.prepare_full(epochs=100, sample_max=sample_max, base_seed=0)
        t3 = time.perf_counter()
        sb.step()
        sb.launch_full()
        t4 = time.perf_counter()
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        res.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
    r = res[-1]
    print(f"J={n_jobs:3d}: graph build {r[0] * 1e3:5.2f}  SweepBatch {r[1] * 1e3:5.2f}  prepare_full {r[2] * 1e3:5.2f}  launch {r[3] * 1e3:5.2f}  device tail {r[4] * 1e3:6.2f} ms")
