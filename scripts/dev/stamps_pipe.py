#!/usr/bin/env python3
"""Diagnostic build only (make STAMPS=1): per-item phase times of the pipelined row-lane SpMM (wave 0 of each
workgroup; the stamps drain wave 0's queues, so the numbers are perturbed).  WDG_SPMM_ABLATE applies."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from wdg_amd import sweep, synth
from wdg_amd._lib import LIB_PATH

lib = ctypes.CDLL(LIB_PATH)
batch = sweep.SweepBatch(sweep.make_jobs(synth.H_LEVELS_10, range(10), k=2), n_feat=500)
assert batch.spmm.plan()[0] == 3
for _ in range(3):
    batch.spmm.launch()
torch.cuda.synchronize()
nb = 256 * 5  # first five items of every workgroup
buf = np.zeros(nb * 16, np.uint64)
assert lib.wdg_debug_rl_stamps(buf.ctypes.data_as(ctypes.c_void_p), nb) == 0
t = buf.reshape(nb, 16).astype(np.float64) * 10e-3  # 100 MHz -> us
names = [("item top -> block 0 ready", 0, 1), ("sweep 0", 1, 2), ("-> block 1 ready", 2, 3), ("sweep 1", 3, 4),
         ("tail: scales, next first chunk", 4, 9), ("pre-epilogue barrier", 9, 10), ("epilogue (stores complete)", 10, 11),
         ("whole item", 0, 11)]
print(f"span of the first 5 items per workgroup {t[:, 11].max() - t[:256, 0].min():.1f} us")
for name, a, b in names:
    d = t[:, b] - t[:, a]
    print(f"   {name:40s} mean {d.mean():7.2f} us  p10 {np.percentile(d, 10):7.2f}  p50 {np.percentile(d, 50):7.2f}  p90 {np.percentile(d, 90):7.2f}")
gap = t[256:, 0] - t[:-256, 11]
print(f"   {'gap between items':40s} mean {gap.mean():7.2f} us")
