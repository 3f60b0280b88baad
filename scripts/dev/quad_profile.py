#!/usr/bin/env python3
"""Where the waves of the quad-row kernel's pipelined loop spend their clocks (dev tool; needs a library built with
QUAD_EXTRA=-DWDG_Q_PROFILE scripts/dev/build_quad_variants.sh ..., passed through WDG_LIB_PATH).
usage: [N=4000] quad_profile.py [k] [seeds]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from wdg_amd import sweep, synth
from wdg_amd._lib import LIB_PATH

k = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
levels = synth.H_LEVELS_10 if k == 2 else synth.H_LEVELS_10_K10
jobs = sweep.make_jobs(levels, range(seeds), k=k, n_nodes=int(os.environ.get("N", 2000)))  # (N=4000: HALF slabs)
lib = ctypes.CDLL(LIB_PATH)
batch = sweep.SweepBatch(jobs, n_feat=500, gcn_hidden=0)
if os.environ.get("TUNE", "1") != "0":
    batch.spmm.tune()
for _ in range(3):
    batch.spmm.launch()
torch.cuda.synchronize()
# padded sweep steps against the ideal: every stored entry of a 16-row slice costs one LDS read per 16-feature group
real = sum(int(g.nnz) for g in batch.graphs)
steps = 0
for g in batch.graphs:
    ext = g.quad["ext"].cpu().numpy().reshape(-1, 2)[: g.quad["n_entries"] * g.quad["n_blocks"]]
    w = ext[:, 1] & 0xffff
    steps += int(((w + 3) // 4 * 4).sum())
print(f"stored entries {real}; sweep steps x 16 rows {steps * 16} = {steps * 16 / real:.3f} x the entries")
n = 1
assert lib.wdg_debug_q_profile(None, 256, 1) == 0
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
batch.spmm.launch()
b.record()
torch.cuda.synchronize()
buf = np.zeros(256 * 16 * 8, np.uint64)
assert lib.wdg_debug_q_profile(buf.ctypes.data_as(ctypes.c_void_p), 256, 0) == 0
r = buf.reshape(256, 16, 8).astype(np.float64)
act = r[:, :, 3] > 0
issue, swp, wait, iters, total = (r[:, :, i][act] for i in range(5))
us = lambda v: v / 100.0  # 100 MHz ticks -> us
print(f"{os.path.basename(LIB_PATH)} k={k} seeds={seeds}: launch {a.elapsed_time(b) * 1e3:.1f} us; waves {act.sum()}, "
      f"iterations per wave {iters.mean():.1f} (min {iters.min():.0f} max {iters.max():.0f}); clock held "
      f"{np.median(r[:, :, 5][act] / r[:, :, 4][act]) * 100:.0f} MHz")
print(f"  per wave, us: loop {us(total).mean():.1f} (min {us(total).min():.1f} max {us(total).max():.1f}); issue {us(issue).mean():.1f}; "
      f"sweep+store {us(swp).mean():.1f}; wait {us(wait).mean():.1f} (max {us(wait).max():.1f})")
print(f"  per iteration, us: issue {us(issue.sum() / iters.sum()):.2f} sweep {us(swp.sum() / iters.sum()):.2f} wait {us(wait.sum() / iters.sum()):.2f}")
t0 = r[:, :, 6][act].min()
begin = np.where(act, r[:, :, 6], np.nan) - t0
end = np.where(act, r[:, :, 7], np.nan) - t0
tot = np.where(act, r[:, :, 4], np.nan)
print("  per XCD (us): first loop start (mean) | last wave end (max) | waves' loop time mean / max | spread inside a workgroup (max - min of its waves' ends, mean)")
for x in range(8):
    sl = slice(x, 256, 8)
    print(f"    xcd {x}: {us(np.nanmean(np.nanmin(begin[sl], axis=1))):6.1f} | {us(np.nanmax(end[sl])):6.1f} | {us(np.nanmean(tot[sl])):6.1f} / {us(np.nanmax(tot[sl])):6.1f} | "
          f"{us(np.nanmean(np.nanmax(end[sl], axis=1) - np.nanmin(end[sl], axis=1))):5.1f}")
