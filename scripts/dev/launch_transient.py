#!/usr/bin/env python3
"""The aggregation launch over the first steps after a synchronisation (dev tool): HIP-event time of the launch next to the
span of its workgroups on the device clock (wdg_spmm_quad_batched_clocked_f32) - is the 110 -> 117 -> 104 us drift inside the
kernel or around it?  usage: launch_transient.py [throttle]   (throttle: keep the host at most 2 steps ahead of the device)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from wdg_amd import sweep, synth
throttle = len(sys.argv) > 1
jobs = sweep.make_jobs(synth.H_LEVELS_10_K10, range(5), k=10)
b = sweep.SweepBatch(jobs, n_feat=500, tune=True)
n = 64
clocks = [b.spmm.new_clock() for _ in range(n)]
for _ in range(300):
    b.step()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
done = [torch.cuda.Event() for _ in range(n)]
for s in range(n):
    ev[s][0].record(); b.spmm.launch(clock=clocks[s]); ev[s][1].record(); b.step_rest(); done[s].record()
    if throttle and s >= 2:
        done[s - 2].synchronize()
torch.cuda.synchronize()
evt = [a.elapsed_time(c) * 1e3 for a, c in ev]
span, start_skew = [], []
for c in clocks:
    t = c.cpu().numpy().reshape(-1, 2).astype(np.float64) * 10e-3  # 100 MHz -> us
    span.append(t[:, 1].max() - t[:, 0].min()); start_skew.append(t[:, 0].max() - t[:, 0].min())
print("throttle" if throttle else "free", "event us :", " ".join(f"{v:.0f}" for v in evt[:48]))
print("            wg span us:", " ".join(f"{v:.0f}" for v in span[:48]))
print("            start skew:", " ".join(f"{v:.0f}" for v in start_skew[:48]))

# the loop of bench.py: every fourth step marked, nothing else
for trial in range(2):
    torch.cuda.synchronize()
    n2 = 60
    ev2 = {s: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for s in range(2, n2, 4)}
    t0 = time.perf_counter()
    for s in range(n2):
        if s in ev2:
            ev2[s][0].record(); b.spmm.launch(); ev2[s][1].record()
        else:
            b.spmm.launch()
        b.step_rest()
    rows = b.results()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("bench-like loop:", " ".join(f"{s}:{a.elapsed_time(c) * 1e3:.0f}" for s, (a, c) in sorted(ev2.items())), f"| {el / n2 * 1e6:.1f} us per step")

# the same loop with every launch clocked: does the workgroups' span follow the event time?
torch.cuda.synchronize()
n3 = 60
clocks3 = [b.spmm.new_clock() for _ in range(n3)]
ev3 = {s: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for s in range(2, n3, 4)}
for _ in range(100):
    b.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(n3):
    if s in ev3:
        ev3[s][0].record(); b.spmm.launch(clock=clocks3[s]); ev3[s][1].record()
    else:
        b.spmm.launch(clock=clocks3[s])
    b.step_rest()
torch.cuda.synchronize()
el = time.perf_counter() - t0
sp3, first3 = [], []
for c in clocks3:
    t = c.cpu().numpy().reshape(-1, 2).astype(np.float64) * 10e-3
    sp3.append(t[:, 1].max() - t[:, 0].min()); first3.append(t[:, 0].min())
gap = [first3[s + 1] - first3[s] for s in range(n3 - 1)]
print("clocked bench-like: events", " ".join(f"{s}:{a.elapsed_time(c) * 1e3:.0f}" for s, (a, c) in sorted(ev3.items())), f"| {el / n3 * 1e6:.1f} us per step")
print("   wg span by step:", " ".join(f"{v:.0f}" for v in sp3))
print("   start-to-start :", " ".join(f"{v:.0f}" for v in gap))
