#!/usr/bin/env python3
"""Where the wall clock of ONE rank's share of the whole sweep goes (dev tool): host times at which every base-shard has been queued /
fetched, and the device times (stream events) at which its launches start and end.   python scripts/dev/rank_phases.py [world] [rank]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from wdg_amd import sweep, synth


class A:
    nodes, kr_epochs = 2000, 100


synth.DUPLICATE_FRACTION = 0.033
world, rank = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 0)
inp = bench.whole_inputs(A)
pairs, graphs, feats = inp["pairs"], inp["graphs"], inp["feats"]
graph_of = lambda j: graphs[(j.h, j.seed)]  # noqa: E731
for _ in range(3):
    sweep.whole_sweep_rank(pairs, graph_of, feats, world, rank)
torch.cuda.synchronize()
log = []
t0 = [0.0]
orig_launch, orig_full, orig_init, orig_finish = sweep.SweepBatch.launch_full, sweep.SweepBatch.full_metrics, sweep.SweepBatch.__init__, None


def launch_full(self, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig_launch(self, *a, **k)
    e1.record()
    log.append(("queued", time.perf_counter() - t0[0], e0, e1, self.n_feat))


def full_metrics(self, *a, **k):
    ta = time.perf_counter() - t0[0]
    r = orig_full(self, *a, **k)
    log.append(("fetched", ta, time.perf_counter() - t0[0], self.n_feat))
    return r


sweep.SweepBatch.launch_full, sweep.SweepBatch.full_metrics = launch_full, full_metrics
for rep in range(3):
    log.clear()
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True)
    start.record()
    t0[0] = time.perf_counter()
    sweep.whole_sweep_rank(pairs, graph_of, feats, world, rank)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0[0]
    print(f"pass {rep}: {total * 1e3:.1f} ms")
    for rec in log:
        if rec[0] == "queued":
            print(f"  F={rec[4]:5d} launches queued at host {rec[1] * 1e3:6.1f} ms | device: {start.elapsed_time(rec[2]):6.1f} -> {start.elapsed_time(rec[3]):6.1f} ms")
        else:
            print(f"  F={rec[3]:5d} fetch: host waits from {rec[1] * 1e3:6.1f} to {rec[2] * 1e3:6.1f} ms")
