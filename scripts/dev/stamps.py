#!/usr/bin/env python3
"""Diagnostic build only (make clean; make STAMPS=1): per-workgroup phase times of the slab SpMM."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from wdg_amd import sweep, synth
from wdg_amd._lib import LIB_PATH

lib = ctypes.CDLL(LIB_PATH)
jobs = sweep.make_jobs(synth.H_LEVELS_10, range(10), k=2)
for slab, thr in ((8, 512), (16, 1024)):
    os.environ["WDG_SPMM_SLAB"], os.environ["WDG_SPMM_THREADS"] = str(slab), str(thr)
    batch = sweep.SweepBatch(jobs, n_feat=500)
    for _ in range(3):
        batch.spmm.launch()
    torch.cuda.synchronize()
    nb = min(8192, 100 * ((500 + slab - 1) // slab))
    buf = np.zeros(nb * 8, np.uint64)
    assert lib.wdg_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p), nb) == 0
    t = buf.reshape(nb, 8).astype(np.float64) * 10e-3  # 100 MHz ticks -> us
    t0 = t[:, 0].min()
    d = np.diff(t[:, :5], axis=1)
    print(f"slab={slab} thr={thr}: blocks={nb}; kernel span {t[:, 4].max() - t0:.1f} us")
    for name, col in (("job fetch + slab/rowptr loads issued+landed (this wave)", 0), ("barrier wait", 1),
                      ("row stream (wave 0)", 2), ("tail barrier (slowest wave)", 3)):
        print(f"   {name:58s} mean {d[:, col].mean():7.2f} us  p10 {np.percentile(d[:, col], 10):7.2f}  p90 {np.percentile(d[:, col], 90):7.2f}")
    life = t[:, 4] - t[:, 0]
    print(f"   workgroup lifetime mean {life.mean():.2f} us; start times: first {0:.1f}, median {np.median(t[:, 0]) - t0:.1f}, last {t[:, 0].max() - t0:.1f} us")
    # by graph density: item -> job -> h level
    per = (500 + slab - 1) // slab
    del batch
