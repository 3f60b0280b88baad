#!/usr/bin/env python3
"""Experiment (dev tool): does parking finished rows in LDS and storing them after the index stream help?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from wdg_amd import sweep, synth

jobs = sweep.make_jobs(synth.H_LEVELS_10, range(10), k=2)
for slab, thr in ((8, 1024), (8, 512), (4, 512), (4, 1024)):
    os.environ["WDG_SPMM_SLAB"], os.environ["WDG_SPMM_THREADS"] = str(slab), str(thr)
    for ab in (0, 8):
        os.environ["WDG_SPMM_ABLATE"] = str(ab)
        try:
            batch = sweep.SweepBatch(jobs, n_feat=500)
            for _ in range(3):
                batch.spmm.launch()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                batch.spmm.launch()
            b.record()
            torch.cuda.synchronize()
            ref = batch.y[3].clone()
            print(f"slab={slab} thr={thr} defer={ab == 8}: {a.elapsed_time(b) / 20 * 1e3:8.1f} us  checksum {float(ref.double().sum()):.6f}", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"slab={slab} thr={thr} defer={ab == 8}: {e}")
