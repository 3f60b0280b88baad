#!/usr/bin/env python3
"""What SweepBatch.tune() (the feedback-balanced tape cut) is worth on the literal N = 4000 / 800 shards and the headline shard, at
several round counts (dev tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "scripts"))
import torch

import bench_configs as bc
from wdg_amd import sweep as sw, synth

for n_nodes, k, seeds in ((4000, 10, 5), (2000, 10, 5)):
    h_levels = synth.H_LEVELS_10 if k == 2 else synth.H_LEVELS_10_K10
    jobs = sw.make_jobs(h_levels, range(seeds), k=k, n_nodes=n_nodes)
    for rounds in (0, 6, 12, 20):
        sb = sw.SweepBatch(jobs, n_feat=500, gcn_hidden=64)
        if rounds:
            sb.tune(rounds=rounds)
        a = bc.timed(sb.spmm.launch, 20)
        print(f"N={n_nodes} k={k}: tune rounds {rounds:2d}: launch mean {a[0]:.1f} us, median {a[1]:.1f}", flush=True)
        del sb
        torch.cuda.empty_cache()
