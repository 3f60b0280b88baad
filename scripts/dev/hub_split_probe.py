#!/usr/bin/env python3
"""Probe (dev tool): the quad-row kernel on the real-topology graphs with HUB ROWS SPLIT into virtual rows of <= L entries (extra rows
behind the graph's own; their partial sums would be added back by a combine pass, not timed here) - what the LDS route could reach
once no wave carries a 700 - 1900-entry slice alone.   WDG_SPMM_BAND=0 python scripts/dev/hub_split_probe.py [L]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from _golden import load
from wdg_amd import ops

L = int(sys.argv[1]) if len(sys.argv) > 1 else 128


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for name, f in (("real_cora", 1433), ("topo_chameleon", 2325), ("topo_squirrel", 2089)):
    g0 = load(name)
    n = int(g0["n_nodes"])
    g = ops.CsrGraph.from_coo(g0["adj_row"], g0["adj_col"], n, None, ops.COO_ADD_SELF_LOOPS)
    rowptr, col = g.rowptr.cpu().numpy().astype(np.int64), g.col.cpu().numpy()
    lens = np.diff(rowptr)
    pieces = np.maximum(1, -(-lens // L))
    src, dst = [], []
    extra_dest = []
    for i in np.flatnonzero(pieces > 1):
        for p in range(1, pieces[i]):
            extra_dest.append(i)
    v = len(extra_dest)
    rows_new = []
    nxt = n
    for i in range(n):
        a, b = rowptr[i], rowptr[i + 1]
        rows_new.append(np.full(min(L, b - a), i))
        for p in range(1, pieces[i]):
            rows_new.append(np.full(min(L, b - a - p * L), nxt))
            nxt += 1
    rows_new = np.concatenate(rows_new)
    g2 = ops.CsrGraph.from_coo(rows_new, col.astype(np.int64), n + v, None, 0)
    g2.n_cols = n  # (same columns: the extra rows only receive)
    rng = np.random.default_rng(17)
    x = torch.from_numpy(((rng.random((n, f), dtype=np.float32) < 0.02) * rng.random((n, f), dtype=np.float32))).cuda()
    d = ops.degree_norm(g, 1, ops.PREC_F32)["dinv"]
    y = torch.empty((n, f), device="cuda")
    t_plain = timed(lambda: ops.spmm(g, x, row_scale=d, col_scale=d, out=y))
    y2 = torch.empty((n + v, f), device="cuda")
    d2 = torch.cat([d, d[torch.tensor(extra_dest, dtype=torch.long, device="cuda")]]) if v else d
    try:
        g2c = ops.CsrGraph(g2.rowptr, g2.col, g2.val, n + v, n)
        t_split = timed(lambda: ops.spmm(g2c, x, row_scale=d2, col_scale=d, out=y2))
        # combine on the host side of the probe (torch): partial sums of the extra rows added to their destination rows, in order
        yc = y2[:n].clone()
        if v:
            yc.index_add_(0, torch.tensor(extra_dest, dtype=torch.long, device="cuda"), y2[n:])
        err = float((yc - y).abs().max() / y.abs().max())
        print(f"{name}: N={n} max row {lens.max()} -> {v} extra rows of <= {L}; as it is {t_plain:.1f} us; split {t_split:.1f} us (+ combine); rel diff {err:.1e}; "
              f"quad={bool(g2c.quad)} split_form={g2c.quad and g2c.quad['split']}")
    except Exception as e:  # noqa: BLE001
        print(name, "failed:", e)
