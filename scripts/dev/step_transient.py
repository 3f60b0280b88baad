"""The aggregation launch, step by step, behind an idle gap (dev tool): after a few ms without work the chip runs steps ~10 - 35 of
the next burst 8 - 12 % slower (power management), which is what a 20-step timed region sees when the host idles before it."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from wdg_amd import sweep, synth
jobs = sweep.make_jobs(synth.H_LEVELS_10_K10, range(5), k=10)
sb = sweep.SweepBatch(jobs, n_feat=500, tune=True)
for _ in range(20): sb.step()
def run(n, label, gap_ms=0.0):
    torch.cuda.synchronize()
    if gap_ms: time.sleep(gap_ms / 1e3)
    ev = []
    for s in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); sb.spmm.launch(); b.record(); sb.step_rest(); ev.append((a, b))
    torch.cuda.synchronize()
    us = [round(a.elapsed_time(b) * 1e3) for a, b in ev]
    print(label, " ".join(str(u) for u in us[::4]))
for gap in (0.0, 0.5, 1.0, 2.0, 5.0, 20.0, 0.0):
    run(80, "idle %4.1f ms before the run, launch us of every 4th step:" % gap, gap)
