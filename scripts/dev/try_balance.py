#!/usr/bin/env python3
"""What SweepBatch.tune() (feedback on the segments' measured spans) does to the C3 launch: spans per XCD and the launch time
round by round (dev tool).  usage: try_balance.py [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from wdg_amd import sweep, synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
jobs = sweep.make_jobs(synth.H_LEVELS_10_K10, range(5), k=10)
sb = sweep.SweepBatch(jobs, n_feat=500)
sp = sb.spmm
clock = sp.new_clock()
shares = np.ones(8)
for rnd in range(rounds):
    sp._set_segments(0, None if rnd == 0 else shares)
    for _ in range(3):
        sb.step()
    spans, ts = np.zeros(8), []
    for _ in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        sp.launch(clock=clock)
        b.record()
        sb.step_rest()
        torch.cuda.synchronize()
        spans += sp.segment_spans(clock) / 8
        ts.append(a.elapsed_time(b) * 1e3)
    print(f"round {rnd}: launch {np.median(ts):6.1f} us; spans " + " ".join(f"{v:5.1f}" for v in spans) + f"  (max - min {spans.max() - spans.min():.1f})", flush=True)
    shares = shares * (spans.mean() / spans) ** 0.7
    shares /= shares.sum() / 8
