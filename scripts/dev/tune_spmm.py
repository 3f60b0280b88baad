#!/usr/bin/env python3
"""Kernel-variant sweep for the batched aggregation (dev tool): times wdg_spmm_batched_f32 on the bench
workload for every (slab, threads) choice through the WDG_SPMM_* overrides.  Usage: python scripts/dev/tune_spmm.py [k]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from wdg_amd import sweep, synth

k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
levels = synth.H_LEVELS_10 if k == 2 else synth.H_LEVELS_10_K10
batch = sweep.SweepBatch(sweep.make_jobs(levels, range(seeds), k=k), n_feat=500)
alg = batch.spmm_algorithmic_bytes()
print(f"k={k} graphs={len(batch.jobs)} edges={batch.edges} alg_bytes={alg/1e6:.1f} MB unique={batch.spmm_unique_bytes()/1e6:.1f} MB")


def timeit(n=30):
    for _ in range(5):
        batch.spmm.launch()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        batch.spmm.launch()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for slab in (4, 8, 16):
    for thr in (512, 1024):
        os.environ["WDG_SPMM_SLAB"], os.environ["WDG_SPMM_THREADS"] = str(slab), str(thr)
        try:
            us = timeit()
            print(f"slab={slab:2d} threads={thr:4d}: {us:8.1f} us  -> {alg/us/1e3:7.1f} GB/s algorithmic ({alg/us/1e3/80:.1f}% of 8 TB/s)")
        except Exception as e:  # noqa: BLE001
            print(f"slab={slab} threads={thr}: {e}")
os.environ.pop("WDG_SPMM_SLAB"), os.environ.pop("WDG_SPMM_THREADS")
os.environ["WDG_SPMM_FORCE_GATHER"] = "1"
us = timeit(10)
print(f"row-gather family: {us:8.1f} us -> {alg/us/1e3:7.1f} GB/s")
