#!/usr/bin/env python3
"""Timing-only ablations of the slab SpMM (dev tool): which phase costs what.  Outputs are WRONG under ablation."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from wdg_amd import ops, sweep, synth

cfgs = [(8, 512), (16, 1024)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
jobs = sweep.make_jobs(synth.H_LEVELS_10, range(10), k=2)
names = {0: "full", 1: "no Y store", 2: "no X load", 4: "no edge loop", 3: "no X load, no Y store", 5: "no store, no edges",
         6: "no load, no edges", 7: "nothing (launch + rowptr)"}
for slab, thr in cfgs:
    os.environ["WDG_SPMM_SLAB"], os.environ["WDG_SPMM_THREADS"] = str(slab), str(thr)
    for ab in (0, 1, 2, 4, 3, 5, 6, 7):
        ops.ABLATE_BITS = ab
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
        print(f"slab={slab} thr={thr} ablate={ab} ({names[ab]:28s}): {a.elapsed_time(b) / 20 * 1e3:8.1f} us", flush=True)
        del batch
