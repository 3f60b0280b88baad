#!/usr/bin/env python3
"""step through tests/test_gpu_kernels.py::test_spmm_quad_family_shapes for one shape, printing progress (dev tool)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["WDG_SPMM_BAND"] = "0"
import numpy as np
import torch

from wdg_amd import ops

from wdg_amd import aggregate
aggregate.ABLATE_BITS = int(os.environ.get("WDG_ABLATE", "0"))
only = os.environ.get("ONLY")

n, m, f, e = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (2000, 2000, 512, 60000)
rng = np.random.default_rng(n * 11 + f)
src, dst = rng.integers(0, n, e), rng.integers(0, n, e)
dst = dst % m
key = np.unique(src * m + dst)
rows, col = (key // m).astype(np.int64), (key % m).astype(np.int32)
rowptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]).astype(np.int32)
val = rng.random(col.shape[0], dtype=np.float32)
x = rng.standard_normal((m, f)).astype(np.float32)
g = ops.CsrGraph(torch.from_numpy(rowptr).cuda(), torch.from_numpy(col).cuda(), torch.from_numpy(val).cuda(), n, m)
assert g.ensure_quad(max_padding=1e9)
torch.cuda.synchronize()
print("quad built", {k: v for k, v in g.quad.items() if not torch.is_tensor(v) and not isinstance(v, np.ndarray)}, flush=True)
d, dc = rng.random(n, dtype=np.float32), rng.random(m, dtype=np.float32)
import scipy.sparse as sp
for name, uv, rs, cs, dt in (("values", True, None, None, torch.float32), ("row scale", False, d, None, torch.float32),
                             ("both scales", False, d, dc, torch.float32), ("bf16", True, d, dc, torch.bfloat16)):
    if only and name != only:
        continue
    xt = torch.from_numpy(x).cuda().to(dt)
    y = ops.spmm(g, xt, row_scale=None if rs is None else torch.from_numpy(rs).cuda(),
                 col_scale=None if cs is None else torch.from_numpy(cs).cuda(), use_values=uv)
    torch.cuda.synchronize()
    v = val.copy() if uv else np.ones_like(val)
    if cs is not None:
        v = v * cs[col]
    a = sp.csr_matrix((v, col, rowptr), shape=(n, m))
    ref = a @ xt.float().cpu().numpy().astype(np.float64)
    if rs is not None:
        ref = ref * rs[:, None]
    err = np.abs(y.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-30)
    print(f"{name}: max err {err:.2e}", flush=True)
