import os, sys, ctypes
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from wdg_amd import ops, _lib
k, n = 500, 64
for jobs, m in ((100, 512), (100, 2000)):
    a = [torch.randn(m, k, device="cuda") for _ in range(jobs)]
    b = [torch.randn(k, n, device="cuda") for _ in range(jobs)]
    c = [torch.empty(m, n, device="cuda") for _ in range(jobs)]
    batch = ops.GemmBatch(list(zip(a, b, c, [None] * jobs)), relu=True)
    for _ in range(5):
        batch.launch()
    torch.cuda.synchronize()
    out = np.zeros(4 * 4096, np.uint64)
    f = ctypes.CDLL(_lib.LIB_PATH).wdg_debug_bres_clocks
    f.argtypes = [ctypes.c_void_p]
    f(out.ctypes.data)
    nw = min(4096, jobs * (1 if m == 512 else 2) * 16)
    o = out.reshape(-1, 4)[:nw].astype(np.int64)
    t0 = o[:, 0].min()
    o = (o - t0) / 100.0
    print(jobs, m, "waves", nw)
    print("  wg start  : min %.1f mean %.1f max %.1f" % (o[:, 0].min(), o[:, 0].mean(), o[:, 0].max()))
    print("  staged    : min %.1f mean %.1f max %.1f" % (o[:, 1].min(), o[:, 1].mean(), o[:, 1].max()))
    print("  last kloop: min %.1f mean %.1f max %.1f" % (o[:, 2].min(), o[:, 2].mean(), o[:, 2].max()))
    print("  end       : min %.1f mean %.1f max %.1f" % (o[:, 3].min(), o[:, 3].mean(), o[:, 3].max()))
    print("  per wave (kloop end - staged) of wg 0:", np.round(o[:16, 2] - o[:16, 1], 1))
    print("  per wave end of wg 0:", np.round(o[:16, 3], 1))
