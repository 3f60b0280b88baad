#!/usr/bin/env python3
"""Diagnostic build only (make clean; make STAMPS=1): per-workgroup phase times of the row-lane SpMM."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from wdg_amd import sweep, synth
from wdg_amd._lib import LIB_PATH

lib = ctypes.CDLL(LIB_PATH)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
levels = synth.H_LEVELS_10 if k == 2 else synth.H_LEVELS_10_K10
batch = sweep.SweepBatch(sweep.make_jobs(levels, range(10), k=k), n_feat=500)
for _ in range(3):
    batch.spmm.launch()
torch.cuda.synchronize()
nb = 1600
buf = np.zeros(nb * 16, np.uint64)
assert lib.wdg_debug_rl_stamps(buf.ctypes.data_as(ctypes.c_void_p), nb) == 0
t = buf.reshape(nb, 16).astype(np.float64) * 10e-3  # 100 MHz -> us
t0 = t[:, 0].min()
names = [("stage blk0 (issue+land+LDS write, wave 0)", 0, 1), ("barrier 0", 1, 2), ("sweep blk0 (wave 0)", 2, 3),
         ("stage blk1 incl. wait for slowest wave", 3, 4), ("barrier 1", 4, 5), ("sweep blk1 (wave 0)", 5, 12),
         ("pre-epilogue barrier", 12, 13), ("epilogue (wave 0)", 13, 14), ("whole workgroup", 0, 14)]
# item -> h level: item = job * 16 + group, job = seed * 10 + level
print(f"k={k}: kernel span {t[:, 14].max() - t0:.1f} us, blocks {nb}")
for name, a, b in names:
    d = t[:, b] - t[:, a]
    print(f"   {name:44s} mean {d.mean():7.2f} us  p10 {np.percentile(d, 10):7.2f}  p50 {np.percentile(d, 50):7.2f}  p90 {np.percentile(d, 90):7.2f}")
