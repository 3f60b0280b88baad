#!/usr/bin/env python3
"""VERDICT r04 item 7, as a bounded experiment: does the step gain from scheduling by SEED GROUP - aggregate a group's graphs,
transform those while their Y is still in L2 / the Infinity Cache, next group - instead of aggregating all 50 graphs (205 MB of Y)
and then transforming all of them?  Same kernels, same job tables per graph: the 50-graph shard is cut into batches of whole
seeds (5 x 10 graphs, 3 + 2 seeds, 2 + 2 + 1), every batch steps in turn on the same streams; per-graph results must be bit for bit
the one-batch step's.   python scripts/dev/ab_chunked_step.py [steps]   (run on the GPU box; prints a table for profiles/)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from wdg_amd import sweep, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
jobs = sweep.make_jobs(synth.H_LEVELS_10_K10, range(5), k=10, n_nodes=2000)


def timed(fn, n):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def build(groups, tune):
    batches = [sweep.SweepBatch([j for j in jobs if j.seed in g], n_feat=500) for g in groups]
    for b in batches:
        b.step()
        if tune:
            b.tune()
    return batches


print(f"{'schedule':34s} {'ms / pass (plain)':>18s} {'ms / pass (one hipGraph)':>26s} {'rows equal':>11s}")
ref_rows = None
for name, groups in (("one batch: 50 graphs", [range(5)]), ("3 + 2 seeds (30 + 20 graphs)", [range(3), range(3, 5)]),
                     ("2 + 2 + 1 seeds", [range(2), range(2, 4), range(4, 5)]), ("5 x 1 seed (10 graphs each)", [[s] for s in range(5)])):
    batches = build(groups, tune=True)

    def one_pass():
        for b in batches:
            b.step()
    plain = timed(one_pass, steps)
    graph = torch.cuda.CUDAGraph()
    one_pass()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        one_pass()
    replayed = timed(graph.replay, steps)
    rows = torch.cat([b.results() for b in batches]).cpu()
    # (job order inside the concatenation = seed-major = the one batch's order)
    if ref_rows is None:
        ref_rows = rows
    print(f"{name:34s} {plain:18.4f} {replayed:26.4f} {str(bool(torch.equal(rows, ref_rows))):>11s}", flush=True)
    del batches, graph
    torch.cuda.empty_cache()
