#!/usr/bin/env python3
"""Timing-only ablations + per-workgroup spans of the quad-row SpMM on the bench workloads (dev tool; outputs are WRONG
under ablation).  reserved bit 0 = stores redirected to rows 0..15 (L2-resident), bit 1 = no X staging, bit 2 = no sweep.
usage: ablate_quad.py [k] [seeds] [codes ...]     (per-workgroup spans need `make clean; make STAMPS=1`)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from wdg_amd import aggregate, ops, sweep, synth  # noqa: F401
from wdg_amd._lib import LIB_PATH

k = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
codes = [int(a) for a in sys.argv[3:]] or [0, 1, 4, 5, 2]
levels = synth.H_LEVELS_10 if k == 2 else synth.H_LEVELS_10_K10
jobs = sweep.make_jobs(levels, range(seeds), k=k, n_nodes=int(os.environ.get("N", 2000)))  # (N=4000: HALF slabs)
names = {0: "full", 1: "stores to L2 only", 2: "no X staging", 4: "no sweep", 5: "pipeline + L2 stores only", 3: "sweep only", 8: "no index stream", 12: "no sweep, no index stream", 13: "no sweep, no index stream, L2 stores"}
lib = ctypes.CDLL(LIB_PATH)
for ab in codes:
    aggregate.ABLATE_BITS = ab
    batch = sweep.SweepBatch(jobs, n_feat=500, gcn_hidden=0)
    for _ in range(3):
        batch.spmm.launch()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        batch.spmm.launch()
    b.record()
    torch.cuda.synchronize()
    line = f"k={k} seeds={seeds} ablate={ab} ({names.get(ab, '?'):26s}): {a.elapsed_time(b) / 20 * 1e3:8.1f} us  segments={getattr(batch.spmm, 'n_segments', None)} items={getattr(batch.spmm, 'n_items', None)}"
    if hasattr(lib, "wdg_debug_q_stamps"):
        buf = np.zeros(256 * 8, np.uint64)
        assert lib.wdg_debug_q_stamps(buf.ctypes.data_as(ctypes.c_void_p), 256) == 0
        t = buf.reshape(256, 8).astype(np.float64) * 10e-3  # 100 MHz -> us
        d = t[:, 1] - t[:, 0]
        per_xcd = [d[x::8].mean() for x in range(8)]
        line += (f"\n     per-workgroup span: mean {d.mean():.1f} min {d.min():.1f} max {d.max():.1f} us; launch span "
                 f"{t[:, 1].max() - t[:, 0].min():.1f}; start skew {t[:, 0].max() - t[:, 0].min():.1f}; per XCD mean "
                 + " ".join(f"{v:.0f}" for v in per_xcd))
    print(line, flush=True)
    del batch
