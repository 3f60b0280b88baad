#!/usr/bin/env python3
"""Cold six-scalar shards per second through sweep.run_shards at several (depth, helper thread) settings (dev tool).
    python scripts/dev/time_cold_modes.py [shards]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from wdg_amd import sweep, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def host_inputs(first_seed):
    jobs = sweep.make_jobs(synth.H_LEVELS_10_K10, range(first_seed, first_seed + 5), k=10, n_nodes=2000)
    feats, inputs = {}, []
    for j in jobs:
        src, dst, lab = synth.regular_graph(j.n_nodes, j.n_classes, j.k, j.h, j.seed)
        feats.setdefault(j.seed, synth.features(j.n_nodes, 500, j.seed))
        inputs.append((src, dst, lab, feats[j.seed]))
    return jobs, inputs


shards = [host_inputs(1000 + 5 * b) for b in range(n)]
for nine in (False, True):
    for depth, force in ((1, "0"), (1, "1"), (2, "0"), (2, "1")):
        os.environ["WDG_SWEEP_BUILD_THREAD_FORCE"] = force
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rows = list(sweep.run_shards(shards, n_feat=500, nine=nine, depth=depth, first_seed=1))
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(f"nine={nine} depth={depth} helper thread={force}: {n * 50 / best:8.0f} graphs/s ({best / n * 1e3:.2f} ms per shard)", flush=True)
