#!/usr/bin/env python3
"""Band kernel against a dense fp64 product on a few shapes + timing vs the other families (diagnostics)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from wdg_amd import ops

def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6

def check(name, g, x, use_values=True):
    n = g.n_rows
    d = ops.degree_norm(g, 1, ops.PREC_F32)["dinv"]
    dense = torch.zeros((g.n_rows, g.n_cols), dtype=torch.float64, device="cuda")
    rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (g.rowptr[1:] - g.rowptr[:-1]).long())
    v = g.val.double() if (use_values and g.val is not None) else torch.ones(g.nnz, dtype=torch.float64, device="cuda")
    dense.index_put_((rows, g.col.long()), v, accumulate=True)
    for rs, cs, tag in ((d, d, "sym"), (d, None, "rw"), (None, None, "raw")):
        want = dense @ (x.double() * (cs.double()[:, None] if cs is not None else 1.0))
        if rs is not None: want = want * rs.double()[:, None]
        os.environ["WDG_SPMM_BAND"] = "1"
        y = ops.spmm(g, x, row_scale=rs, col_scale=cs, use_values=use_values)
        torch.cuda.synchronize()
        err = float((y.double() - want).abs().max() / max(float(want.abs().max()), 1e-30))
        tb = timed(lambda: ops.spmm(g, x, row_scale=rs, col_scale=cs, use_values=use_values, out=y))
        os.environ["WDG_SPMM_BAND"] = "0"
        y2 = ops.spmm(g, x, row_scale=rs, col_scale=cs, use_values=use_values)
        err2 = float((y2.double() - want).abs().max() / max(float(want.abs().max()), 1e-30))
        to = timed(lambda: ops.spmm(g, x, row_scale=rs, col_scale=cs, use_values=use_values, out=y2))
        print(f"{name:28s} {tag:4s} band err {err:.2e} {tb:8.1f} us | other err {err2:.2e} {to:8.1f} us  (hub rows {g.band['n_hub'] if g.band else None})", flush=True)

from _golden import load
rng = np.random.default_rng(1)
for name, f in (("squirrel", 2089), ("chameleon", 2325)):
    g0 = load("topo_" + name); n = int(g0["n_nodes"])
    g = ops.CsrGraph.from_coo(g0["adj_row"], g0["adj_col"], n, None, ops.COO_ADD_SELF_LOOPS)
    x = torch.from_numpy(rng.standard_normal((n, f), dtype=np.float32)).cuda()
    check(name + f" F={f}", g, x)
    check(name + " F=512", g, x[:, :512].contiguous())
    check(name + " F=100 (strided)", g, x[:, 3:103])
d = load("real_cora"); n = int(d["n_nodes"])
g = ops.CsrGraph.from_coo(d["adj_row"], d["adj_col"], n, None, ops.COO_ADD_SELF_LOOPS)
x = torch.from_numpy(rng.standard_normal((n, 1433), dtype=np.float32)).cuda()
check("cora F=1433", g, x)
from wdg_amd import synth
for k, h in ((2, 0.5), (10, 0.15)):
    src, dst, labels = synth.regular_graph(2000, 5, k, h, 1)
    g = ops.CsrGraph.from_coo(src, dst, 2000, None, ops.COO_ADD_SELF_LOOPS)
    x = torch.from_numpy(synth.features(2000, 500, 1)).cuda()
    check(f"regular N=2000 k={k} F=500", g, x)
# a large pattern (bucket sort of the plan), F = 64 and values
src, dst = synth.random_graph(40000, 400000, seed=3, power_law=True)
g = ops.CsrGraph.from_coo(src, dst, 40000, torch.rand(len(src)), ops.COO_SYMMETRISE | ops.COO_ADD_SELF_LOOPS)
x = torch.from_numpy(rng.standard_normal((40000, 64), dtype=np.float32)).cuda()
check("power law N=40000 F=64 val", g, x)
check("power law N=40000 F=20 val", g, x[:, :20])
