#!/usr/bin/env python3
"""Line-level host time of SweepBatch.__init__ / prepare_full inside the whole sweep (dev tool; sys.settrace on those two frames)."""
import os
import sys
import time
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from wdg_amd import sweep


class A:
    nodes, kr_epochs = 2000, 100


inp = bench.whole_inputs(A)
pairs, graphs, feats = inp["pairs"], inp["graphs"], inp["feats"]
graph_of = lambda j: graphs[(j.h, j.seed)]  # noqa: E731
sweep.whole_sweep_rank(pairs[:4], graph_of, feats, 1, 0)
sweep.whole_sweep_rank(pairs, graph_of, feats, 1, 0)
torch.cuda.synchronize()
acc = defaultdict(float)
last = {}
targets = {sweep.SweepBatch.__init__.__code__, sweep.SweepBatch.prepare_full.__code__}


def tracer(frame, event, arg):
    if frame.f_code not in targets:
        return None

    def local(frame, event, arg):
        now = time.perf_counter()
        key = id(frame)
        if key in last:
            code, line, t0 = last[key]
            acc[(code, line)] += now - t0
        if event == "return":
            last.pop(key, None)
        else:
            last[key] = (frame.f_code.co_name, frame.f_lineno, time.perf_counter())
        return local
    last[id(frame)] = (frame.f_code.co_name, frame.f_lineno, time.perf_counter())
    return local


W, R = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 0)
sweep.whole_sweep_rank(pairs, graph_of, feats, W, R)
torch.cuda.synchronize()
sys.settrace(tracer)
t0 = time.perf_counter()
sweep.whole_sweep_rank(pairs, graph_of, feats, W, R)
torch.cuda.synchronize()
sys.settrace(None)
print(f"traced pass: {(time.perf_counter() - t0) * 1e3:.0f} ms")
src = open(sweep.__file__).read().splitlines()
for (code, line), t in sorted(acc.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{t * 1e3:8.1f} ms  {code}:{line}  {src[line - 1].strip()[:110]}")
