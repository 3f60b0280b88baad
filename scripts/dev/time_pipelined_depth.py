#!/usr/bin/env python3
"""sweep.run_shards over 8 distinct 50-graph shards (nine scalars) at pipeline depth 1, 2, 3 (dev tool)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from wdg_amd import sweep, synth


def host_inputs(first_seed):
    jobs = sweep.make_jobs(synth.H_LEVELS_10_K10, range(first_seed, first_seed + 5), k=10, n_nodes=2000)
    feats, inputs = {}, []
    for j in jobs:
        src, dst, lab = synth.regular_graph(j.n_nodes, j.n_classes, j.k, j.h, j.seed)
        feats.setdefault(j.seed, synth.features(j.n_nodes, 500, j.seed))
        inputs.append((src, dst, lab, feats[j.seed]))
    return jobs, inputs


shards = [host_inputs(2000 + 5 * b) for b in range(9)]
list(sweep.run_shards(shards[:1], nine=True, depth=1))  # warm-up
for nine in (True, False):
    for depth in (1, 2, 3, 2, 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rows = list(sweep.run_shards(shards[1:], nine=nine, depth=depth, first_seed=1))
        dt = time.perf_counter() - t0
        print(f"nine={nine} depth={depth}: {dt * 1e3 / 8:.2f} ms per shard = {8 * 50 / dt:.0f} graphs/s", flush=True)
