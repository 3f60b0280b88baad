import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from wdg_amd import ops
n, f, groups = 2000, 512, (10, 10)
rng = np.random.default_rng(n * 3 + f)
entries, want = [], []
for gi, size in enumerate(groups):
    x = torch.from_numpy(rng.standard_normal((n, f)).astype(np.float32)).cuda()
    for j in range(size):
        e = int(n * (2 + 3 * j))
        src, dst = rng.integers(0, n, e), rng.integers(0, n, e)
        g = ops.CsrGraph.from_coo(src, dst, n, rng.random(e, dtype=np.float32), ops.COO_ADD_SELF_LOOPS)
        d = ops.degree_norm(g, ops.NORM_RW)["dinv"]
        want.append(ops.spmm(g, x, row_scale=d, use_values=False).clone())
        entries.append((g, x, torch.full_like(want[-1], float("nan")), d, None, False))
batch = ops.SpmmBatch(entries)
print("items", batch.n_items, "segs", batch.n_segments)
import ctypes
from wdg_amd._lib import SpmmItem
raw = bytes(batch.items.cpu().numpy())
its = (SpmmItem * batch.n_items).from_buffer_copy(raw)
for it in its: print(it.first_job, it.n_jobs, it.unit_begin, it.unit_end)
print(batch.seg_ptr.cpu().tolist())
batch.launch(); torch.cuda.synchronize()
for k, ((g, _, y, _, _, _), w) in enumerate(zip(entries, want)):
    bad = (y != w) & ~(torch.isnan(y) & torch.isnan(w))
    if bad.any():
        rows = bad.any(1).nonzero().flatten().cpu().numpy()
        cols = bad.any(0).nonzero().flatten().cpu().numpy()
        perm = g.quad["perm"].cpu().numpy()
        slot = {int(r): i for i, r in enumerate(perm[:n])}
        slots = sorted(slot[int(r)] for r in rows)
        print("entry", k, "bad rows", len(rows), "slots", slots[:10], "...", slots[-5:], "sus", sorted(set(s // 64 for s in slots)), "cols", cols[:5], len(cols), "nan", int(torch.isnan(y).sum()),
              "maxdiff", float((y - w)[bad].abs().max()))
