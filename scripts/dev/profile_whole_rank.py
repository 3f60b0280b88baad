#!/usr/bin/env python3
"""Where one rank's share of the reference's whole sweep goes on the HOST (dev tool): cProfile of sweep.whole_sweep_rank for
rank 0 of 8 (35 adjacencies x 6 bases), after a warm-up pass.   python scripts/dev/profile_whole_rank.py [world] [rank] [adjacencies per shard]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from wdg_amd import sweep


class A:
    nodes, kr_epochs = 2000, 100


world, rank = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 0)
per_shard = int(sys.argv[3]) if len(sys.argv) > 3 else 80  # adjacencies per shard
inp = bench.whole_inputs(A)
pairs, graphs, feats = inp["pairs"], inp["graphs"], inp["feats"]
graph_of = lambda j: graphs[(j.h, j.seed)]  # noqa: E731
sweep.whole_sweep_rank(pairs[:4], graph_of, feats, 1, 0)
torch.cuda.synchronize()
import gc
for rep in range(4):
    if rep == 2:
        gc.disable()
        print("-- cyclic collector off --")
    t0 = time.perf_counter()
    keys, rows = sweep.whole_sweep_rank(pairs, graph_of, feats, world, rank, max_pairs_per_shard=per_shard)
    torch.cuda.synchronize()
    print(f"rank {rank} of {world}: {keys.shape[0]} rows in {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
pr = cProfile.Profile()
pr.enable()
keys, rows = sweep.whole_sweep_rank(pairs, graph_of, feats, world, rank, max_pairs_per_shard=per_shard)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(30)
if os.environ.get("WDG_PROF_CALLEES"):
    st.sort_stats("cumulative")
    for pat in ("prepare_full", "sweep.py:.*(__init__)", "rebind_features", "kernel_regression.py:.*(__init__)", "aggregate.py:.*(__init__)"):
        st.print_callees(pat)
