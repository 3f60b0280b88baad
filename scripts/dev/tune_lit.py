#!/usr/bin/env python3
"""The literal N = 4000 shard's tuned aggregation launch under tuner settings (dev tool): rounds / finalists A/B on one box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wdg_amd import sweep, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
jobs = sweep.make_jobs(synth.H_LEVELS_10_K10, range(5), k=10, n_nodes=n)
def launch_us(sb, reps=40):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); sb.spmm.launch(); b.record(); sb.step_rest(); ts.append((a, b))
    torch.cuda.synchronize()
    us = sorted(a.elapsed_time(b) for a, b in ts)
    return us[len(us) // 2] * 1e3
for rep in range(2):
    for name, kw in (("old 6/1", dict(rounds=6, finalists=1)), ("new 10/3", dict()), ("16/4", dict(rounds=16, finalists=4))):
        sb = sweep.SweepBatch(jobs, n_feat=500)
        sb.step(); torch.cuda.synchronize()
        un = launch_us(sb)
        best = sb.tune(**kw)
        print(f"N={n} {name}: untuned {un:.1f} us, tuned {launch_us(sb):.1f} us, burst {({k: round(v*1e3,1) for k,v in best[3].items()}) if best and len(best)>3 else None}, round medians {[round(c*1e3,1) for c in getattr(sb.spmm,'_dbg',[])]}", flush=True)
        del sb
