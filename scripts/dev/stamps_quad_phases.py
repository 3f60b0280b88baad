#!/usr/bin/env python3
"""Per-XCD timeline of the quad-row kernel's phases (needs `make STAMPS=1`): staging and sweep time of every phase next to
the super-units and entry widths it holds.  usage: stamps_quad_phases.py [k] [seeds]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from wdg_amd import sweep, synth
from wdg_amd._lib import LIB_PATH
k = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lib = ctypes.CDLL(LIB_PATH)
levels = synth.H_LEVELS_10 if k == 2 else synth.H_LEVELS_10_K10
b = sweep.SweepBatch(sweep.make_jobs(levels, range(seeds), k=k), n_feat=500, gcn_hidden=0)
sp = b.spmm
acc = None
reps = 8
for it in range(3 + reps):
    # (the stamp buffer keeps old values: zero it through a launch-free path is not exported; phases are recognised by order)
    sp.launch(); torch.cuda.synchronize()
    buf = np.zeros(256 * 8, np.uint64)
    assert lib.wdg_debug_q_stamps(buf.ctypes.data_as(ctypes.c_void_p), 256) == 0
    t = buf.reshape(256, 8).astype(np.float64) * 10e-3
    t = t - t[:, :1]
    if it >= 3:
        acc = t if acc is None else acc + t
t = acc / reps
ent, order = sp.keep, sp.order
for x in range(8):
    items = sp.items_host[sp.seg_ptr_host[x]:sp.seg_ptr_host[x + 1]]
    rows = t[x::8]
    line = f"XCD {x}: total {rows[:, 1].mean():6.1f} us |"
    for pi, (fj, nj, ub, ue) in enumerate(items[:3]):
        w = np.concatenate([ent[order[j]][0].quad["widths"].sum(0) for j in range(fj, fj + nj)])[ub * 4:ue * 4]
        start, staged = rows[:, 2 + 2 * pi].mean(), rows[:, 3 + 2 * pi].mean()
        end = rows[:, 4 + 2 * pi].mean() if pi + 1 < len(items) else rows[:, 1].mean()
        line += (f" ph{pi}: {ue - ub:3d} SU w{w[w > 0].mean():4.1f} at {start:5.1f} stage {staged - start:4.1f} sweep {end - staged:5.1f}"
                 f" ({(end - staged) / max(ue - ub, 1) * 16:4.2f}/SU) |")
    print(line, flush=True)
