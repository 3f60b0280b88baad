"""HALF slabs (graphs of 2529 .. 5056 columns, 32-byte slab rows): launch time of the literal C3 shard (N = 4000) and the LDS
cycles per ds_read_b64 service group (rows 0-7 / 8-15: eight 32-byte rows, bank window 8 (column mod 8)) and sweep step of
its SELL-16 copies, for the entry order WDG_SELL_ORDER of the environment.  usage: WDG_SELL_ORDER=0|1|2 python scripts/dev/half_slab.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import wdg_amd  # noqa: E402,F401
from wdg_amd import sweep as sw, synth  # noqa: E402

CONT = 1 << 30
jobs = sw.make_jobs(synth.H_LEVELS_10_K10, range(5), k=10, n_nodes=int(os.environ.get("N", 4000)))
sb = sw.SweepBatch(jobs, n_feat=500, gcn_hidden=0)
for _ in range(5):
    sb.spmm.launch()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
for a, b in ev:
    a.record()
    sb.spmm.launch()
    b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)
print("order", os.environ.get("WDG_SELL_ORDER", "default"), "kernel", sb.spmm.kernel_name(), "launch us mean %.1f median %.1f" % (sum(ms) / len(ms) * 1e3, ms[len(ms) // 2] * 1e3))
tot_c = tot_s = 0
for gi in (0, 3, 6, 9):
    q = sb.graphs[gi].quad
    rb = 32 if q["half"] else 64
    ext, qc = q["ext"].cpu().numpy().reshape(-1, 2), q["col"].cpu().numpy()
    groups = ((0, 1, 2, 3, 4, 5, 6, 7), (8, 9, 10, 11, 12, 13, 14, 15)) if q["half"] else ((0, 3, 5, 6), (1, 2, 4, 7), (8, 11, 13, 14), (9, 10, 12, 15))
    ncls = 8 if q["half"] else 4
    cycles = steps = 0
    for e_i in range(q["n_entries"]):
        c0, word = ext[e_i]
        w = -(-(word & 0xffff) // 4) * 4
        if w == 0:
            continue
        n_chunks = -(-w // 16)
        blk = qc[c0 * 256:(c0 + n_chunks) * 256].reshape(n_chunks, 16, 16).transpose(1, 0, 2).reshape(16, -1)[:, :w]
        for grp in groups:
            rows = blk[list(grp)] // rb  # [rows of the group, steps]
            for e in range(w):
                cycles += np.bincount(np.unique(rows[:, e]) % ncls, minlength=ncls).max()
            steps += w
    print("graph", gi, "h", jobs[gi].h, "cycles per group step %.3f" % (cycles / steps))
    tot_c, tot_s = tot_c + cycles, tot_s + steps
print("all: %.3f" % (tot_c / tot_s))
