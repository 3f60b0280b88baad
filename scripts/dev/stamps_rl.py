#!/usr/bin/env python3
"""Diagnostic build only (make clean; make STAMPS=1): per-item phase times of the persistent row-lane SpMM (wave 0 of
each workgroup; the stamps drain wave 0's queues, so the numbers are perturbed).  WDG_SPMM_ABLATE applies."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
nb = 256 * 6  # first six items of every workgroup
buf = np.zeros(nb * 16, np.uint64)
assert lib.wdg_debug_rl_stamps(buf.ctypes.data_as(ctypes.c_void_p), nb) == 0
t = buf.reshape(nb, 16).astype(np.float64) * 10e-3  # 100 MHz -> us
names = [("descriptor, extents, first chunk (blk 0)", 0, 1), ("stage blk0", 1, 2), ("barrier", 2, 3), ("sweep blk0", 3, 4),
         ("extents, first chunk, out rows (blk 1)", 4, 5), ("stage blk1 (incl. barrier before)", 5, 6), ("barrier", 6, 7),
         ("sweep blk1", 7, 8), ("pre-epilogue barrier", 8, 13), ("epilogue (stores complete)", 13, 14),
         ("final barrier", 14, 15), ("whole item", 0, 15)]
span = t[:, 15].max() - t[:256, 0].min()
print(f"k={k}: span of the first 6 items per workgroup {span:.1f} us")
for name, a, b in names:
    d = t[:, b] - t[:, a]
    print(f"   {name:44s} mean {d.mean():7.2f} us  p10 {np.percentile(d, 10):7.2f}  p50 {np.percentile(d, 50):7.2f}  p90 {np.percentile(d, 90):7.2f}")
gap = t[256:, 0] - t[:-256, 15]
print(f"   {'gap between items':44s} mean {gap.mean():7.2f} us")
