#!/usr/bin/env python3
"""Race hunt (dev tool): launch every row-lane SpMM variant many times on the same inputs and compare each result
bitwise with the default variant's first result.  usage: stress_spmm.py [launches per variant] [seeds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from wdg_amd import sweep, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
first = None
for name, env in (("default", {"WDG_SPMM_RUN": "0"}), ("shared run 5", {"WDG_SPMM_RUN": "5"}),
                  ("shared run 2", {"WDG_SPMM_RUN": "2"}), ("pipelined", {"WDG_SPMM_RUN": "0", "WDG_SPMM_PIPELINED": "1"})):
    os.environ.update(env)
    sb = sweep.SweepBatch(sweep.make_jobs(synth.H_LEVELS_10, range(seeds), k=2), n_feat=500, gcn_hidden=0)
    plan = sb.spmm.plan()
    bad = 0
    for it in range(reps + 1):
        for y in sb.y:
            y.fill_(float("nan"))
        sb.spmm.launch()
        torch.cuda.synchronize()
        if first is None:
            first = [y.clone() for y in sb.y]
        for gi, (y, r) in enumerate(zip(sb.y, first)):
            if not torch.equal(y, r):
                bad += 1
                if bad <= 3:
                    d = (y != r) | torch.isnan(y)
                    rows, cols = d.any(1).nonzero().flatten(), d.any(0).nonzero().flatten()
                    print(f"   {name}: launch {it}, graph {gi}: {int(d.sum())} wrong values, rows {rows[:6].tolist()}.. "
                          f"({rows.numel()}), cols {cols[:6].tolist()}.. ({cols.numel()})", flush=True)
                break
    print(f"{name}: {bad} bad launches of {reps + 1}  plan={plan}", flush=True)
    for k in env:
        os.environ.pop(k)
