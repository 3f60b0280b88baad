#!/usr/bin/env python3
"""The whole sweep on one GPU at pipeline depths 2 / 3 / 4 (dev tool)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from wdg_amd import sweep, synth


class A:
    nodes, kr_epochs = 2000, 100


synth.DUPLICATE_FRACTION = 0.033
inp = bench.whole_inputs(A)
pairs, graphs, feats = inp["pairs"], inp["graphs"], inp["feats"]
graph_of = lambda j: graphs[(j.h, j.seed)]  # noqa: E731
per = int(sys.argv[1]) if len(sys.argv) > 1 else 56
for depth in (2, 3, 4, 2):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sweep.whole_sweep_rank(pairs, graph_of, feats, 1, 0, max_pairs_per_shard=per, depth=depth)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"depth {depth}, {per} adjacencies per shard: {dt * 1e3:.1f} ms (third pass)", flush=True)
