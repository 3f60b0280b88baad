#!/usr/bin/env python3
"""Batched train + eval of one model per graph over the bench shard (100 graphs, N=2000, F=500): epochs/s and graphs/s,
eager launches vs the captured hipGraph (dev tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from wdg_amd import sweep, synth

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
jobs = sweep.make_jobs(synth.H_LEVELS_10, range(10), k=2)
sb = sweep.SweepBatch(jobs, n_feat=500, gcn_hidden=0)
for s in sb.x:
    lab = synth.regular_graph(2000, 5, 2, 0.5, s)[2]
    sb.x[s].copy_(torch.from_numpy(synth.features(2000, 500, s, labels=lab)))
for kind in ("sgc", "gcn"):
    for capture in (False, True):
        tb = sweep.TrainBatch(sb, kind=kind, hidden=64, seed=1)
        r = tb.run(epochs=epochs, capture=capture)
        acc = r["test_acc"].view(10, 10).mean(0)  # per h level (jobs are seed-major)
        print(f"{kind} {'graph ' if capture else 'eager '}: {r['seconds'] / epochs * 1e3:7.3f} ms per epoch for {tb.J} graphs "
              f"-> {tb.J / (r['seconds'] / epochs * 200):8.1f} graphs/s at 200 epochs; test acc by h "
              + " ".join(f"{a:.2f}" for a in acc.tolist()), flush=True)
        del tb
