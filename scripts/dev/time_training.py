#!/usr/bin/env python3
"""Train + eval epoch time of SGC-1 / GCN-2 on a sweep graph, eager vs captured hipGraphs (dev tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from wdg_amd import models, ops, synth

n, f, c, epochs = 2000, 500, 5, 200
for h in (0.1, 0.5, 0.9):
    src, dst, labels = synth.regular_graph(n, c, 2, h, seed=0)
    x = torch.from_numpy(synth.features(n, f, seed=0, labels=labels)).cuda()
    adj = models.NormAdj(ops.CsrGraph.from_coo(src, dst, n, None, ops.COO_ADD_SELF_LOOPS), add_self_loops=False)
    y = torch.from_numpy(labels)
    torch.manual_seed(1)
    masks = models.random_disassortative_splits(y, y.max() + 1)
    for kind in ("sgc", "gcn"):
        out = {}
        for capture in (False, True):
            torch.manual_seed(2)
            m = models.SGC1(f, c) if kind == "sgc" else models.GCN2(f, c, nhid=64, dropout=0.5)
            out[capture] = models.train_eval_graphed(m, adj, x, y, masks=masks, epochs=epochs, capture=capture)
        e, g = out[False], out[True]
        print(f"h={h:4.2f} {kind}: eager {e['seconds'] / epochs * 1e3:6.3f} ms/epoch, graphed {g['seconds'] / epochs * 1e3:6.3f} ms/epoch "
              f"({e['seconds'] / g['seconds']:.1f}x) -> {1.0 / g['seconds']:.1f} graphs/s at {epochs} epochs; test acc {g['test_acc']:.3f}", flush=True)
