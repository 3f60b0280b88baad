#!/usr/bin/env python3
"""The aggregated features' kernels by both routes (dev tool): ops.GramBatch over Y = A_hat X (direct) against ops.PropagatedGram
(A_hat K(X) A_hat^T) for a 70-graph base-shard at the reference's feature widths.   python scripts/dev/time_gram_routes.py [graphs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from wdg_amd import ops, sweep, synth

G = int(sys.argv[1]) if len(sys.argv) > 1 else 70
levels = [h for h in synth.H_LEVELS_30 if h not in (0.05, 0.1)]
jobs = (sweep.make_jobs(levels, range(3), k=10))[:G]
graphs, dinv = [], []
for j in jobs:
    src, dst, _ = synth.regular_graph(j.n_nodes, 5, j.k, j.h, j.seed)
    g = ops.CsrGraph.from_coo(src, dst, j.n_nodes, None, ops.COO_ADD_SELF_LOOPS)
    g.ensure_quad()
    graphs.append(g)
    dinv.append(ops.degree_norm(g, ops.NORM_RW, ops.PREC_F32)["dinv"])


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print(f"{G} graphs (N = 2000, k = 10, the 4000 set's levels), kernels of the aggregated features: linear + arc-cosine")
for f in (500, 932, 1433, 2089, 2325, 3703):
    xs = {s: torch.from_numpy(synth.features(2000, f, s)).cuda() for s in {j.seed for j in jobs}}
    ys = [ops.spmm(g, xs[j.seed], row_scale=d) for j, g, d in zip(jobs, graphs, dinv)]
    direct = ops.GramBatch(ys)
    gx = ops.GramBatch(list(xs.values()))
    kx = dict(zip(xs, gx.k_linear))
    prop = ops.PropagatedGram([(g, d, None, kx[j.seed]) for j, g, d in zip(jobs, graphs, dinv)])
    gx.launch()
    t_d, t_p, t_x = timed(direct.launch), timed(prop.launch), timed(gx.launch)
    parts = {"first": timed(prop.first.launch), "second": timed(prop.second.launch)}
    err = max(float((a - b).abs().max() / a.abs().max()) for a, b in zip(direct.k_linear[:4], prop.k_linear[:4]))
    print(f"F = {f:5d}: direct {t_d:8.2f} ms   propagated {t_p:8.2f} ms (aggregations {parts['first']:.2f} + {parts['second']:.2f})   "
          f"raw features' Grams ({len(xs)}) {t_x:6.2f} ms   max |dK| / max |K| {err:.1e}", flush=True)
    del direct, prop, gx, ys, xs
    torch.cuda.empty_cache()
