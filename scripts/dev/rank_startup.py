#!/usr/bin/env python3
"""The start-up of one rank's share of the whole sweep, step by step on the host clock (dev tool): what runs before the first
base-shard's launches are queued.   python scripts/dev/rank_startup.py [world] [rank]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from wdg_amd import ops, sweep, synth


class A:
    nodes, kr_epochs = 2000, 100


synth.DUPLICATE_FRACTION = 0.033
world, rank = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 0)
inp = bench.whole_inputs(A)
pairs, graphs, feats = inp["pairs"], inp["graphs"], inp["feats"]
graph_of = lambda j: graphs[(j.h, j.seed)]  # noqa: E731
for _ in range(2):
    sweep.whole_sweep_rank(pairs, graph_of, feats, world, rank)
torch.cuda.synchronize()
mine = sweep.pairs_of_rank(pairs, world, rank)
gi = [graph_of(j) for j in mine]
import cProfile
import pstats
for rep in range(3):
    torch.cuda.synchronize()
    marks = [("start", time.perf_counter())]
    pr = cProfile.Profile() if rep == 2 and os.environ.get("WDG_PROF") else None

    def mark(name, sync=False):
        if sync:
            torch.cuda.synchronize()
        marks.append((name, time.perf_counter()))

    gb = ops.GraphBatch([(src, dst, j.n_nodes) for j, (src, dst, _l) in zip(mine, gi)], ops.COO_ADD_SELF_LOOPS, quad=True, defer=True)
    mark("graph build queued (host pack + upload + launches)")
    gb.finish()
    mark("graph build finished (read-back)")
    for bi in (0, 3):
        name, fx, sample_max = feats[bi]
        width = next(iter(fx.values())).shape[1]
        inputs = [(src, dst, lab, fx[j.seed]) for j, (src, dst, lab) in zip(mine, gi)]
        if pr and bi == 0:
            pr.enable()
        sb = sweep.SweepBatch(mine, n_feat=width, gcn_hidden=0, inputs=inputs, labels_only=True, graph_batch=gb if bi == 0 else None,
                              share=None if bi == 0 else first)
        mark(f"SweepBatch {name} F={width} (feature upload, step tables)")
        sb.prepare_full(epochs=100, sample_max=sample_max, base_seed=1000 * bi)
        if pr and bi == 0:
            pr.disable()
        mark("prepare_full (Gram / propagation / set / regression tables)")
        sb.step()
        sb.launch_full()
        mark("step + launch_full queued")
        if bi == 0:
            first = sb
    mark("device done", sync=True)
    print(f"pass {rep}")
    for (_n0, a), (n1, b) in zip(marks, marks[1:]):
        print(f"  {1e3 * (b - a):6.2f} ms  {n1}   (at {1e3 * (b - marks[0][1]):.1f})")
    if pr:
        pstats.Stats(pr).sort_stats("tottime").print_stats(28)
