#!/usr/bin/env python3
"""Does a second workgroup per CU pay for the quad-row aggregation?  (dev tool)

The slab of a 2000-node graph (128 KB) leaves room for one workgroup per CU.  Graphs of 1000 nodes (64-KB slabs) leave room
for two, so the SAME kernel built with 512-thread workgroups, two per CU (`make EXTRA="-DWDG_Q_THREADS=512
-DWDG_Q_WGS_PER_CU=2"`), measures what overlapping one workgroup's staging / barriers / store drain with the other's sweep
is worth before any 8-feature-slab kernel is written.  usage: try_two_wg.py <n_nodes> <n_seeds> [subs ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

n_nodes, seeds = int(sys.argv[1]), int(sys.argv[2])
for subs in [int(a) for a in sys.argv[3:]] or [1, 2]:
    os.environ["WDG_QUAD_SUBS"] = str(subs)
    from wdg_amd import sweep, synth
    jobs = sweep.make_jobs(synth.H_LEVELS_10_K10, range(seeds), k=10, n_nodes=n_nodes)
    batch = sweep.SweepBatch(jobs, n_feat=500, gcn_hidden=0)
    for _ in range(5):
        batch.spmm.launch()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            batch.spmm.launch()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 20 * 1e3)
    edges = sum(int(g.nnz) for g in batch.graphs) if hasattr(batch, "graphs") else -1
    print(f"N={n_nodes} graphs={len(jobs)} subs={subs} segments={batch.spmm.n_segments} items={batch.spmm.n_items}: {best:8.1f} us per launch", flush=True)
    del batch
