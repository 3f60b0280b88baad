#!/usr/bin/env python3
"""One sweep step as plain launches against one captured hipGraph per step (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wdg_amd import sweep, synth
jobs = sweep.make_jobs(synth.H_LEVELS_10_K10, range(5), k=10)
b = sweep.SweepBatch(jobs, n_feat=500, tune=True)
for _ in range(20):
    b.step()
torch.cuda.synchronize()
def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
print("plain launches      :", round(timeit(b.step), 1), "us per step")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): b.step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    b.step()
print("one graph per step  :", round(timeit(g.replay), 1), "us per step")
g4 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g4):
    for _ in range(4): b.step()
print("one graph per 4 steps:", round(timeit(g4.replay, 50) / 4, 1), "us per step")
