import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from _golden import load, dense_features
from wdg_amd import ops
from wdg_amd.utils import homophily_metrics as hm, util_funcs as uf
g0 = load("real_texas")
n = int(g0["n_nodes"])
idx = torch.from_numpy(np.vstack([g0["adj_row"], g0["adj_col"]]).astype(np.int64))
adj = torch.sparse_coo_tensor(idx, torch.ones(idx.shape[1]), (n, n))
x = torch.from_numpy(dense_features(g0)).cuda()
lab = torch.from_numpy(g0["labels"])
g = hm._graph(adj)
h = ops.spmm(g, x)
K = hm._gram_kernel(h, 1).cpu().numpy().astype(np.float64)
torch.manual_seed(11)
sets = uf.kernel_regression_epoch_indices(lab, 200.0, 6)
c = int(lab.max()) + 1
for e, (tr, va) in enumerate(sets):
    tr, va = tr.numpy(), va.numpy()
    Ktt = K[np.ix_(tr, tr)]
    Y = np.eye(c)[lab[tr].numpy()]
    a_pinv32 = np.linalg.pinv(Ktt.astype(np.float32)) @ Y.astype(np.float32)
    a_pinv64 = np.linalg.pinv(Ktt) @ Y
    ev = np.linalg.eigvalsh(Ktt)
    acc = lambda a: float(((K[np.ix_(va, tr)] @ a).argmax(1) == lab[va].numpy()).mean())
    # fp32 cholesky with drop
    A = Ktt.astype(np.float32).copy(); nt = len(tr); L = np.zeros_like(A); drop = np.zeros(nt, bool)
    thr = nt * 1.19e-7 * A.diagonal().max()
    for k in range(nt):
        piv = A[k, k]
        if not piv > thr:
            drop[k] = True; L[k, k] = 1; continue
        L[k, k] = np.sqrt(piv); L[k+1:, k] = A[k+1:, k] / L[k, k]
        A[k+1:, k+1:] -= np.outer(L[k+1:, k], L[k+1:, k])
    Yd = Y.astype(np.float32).copy(); Yd[drop] = 0
    y = np.zeros_like(Yd)
    for k in range(nt):
        y[k] = 0 if drop[k] else (Yd[k] - L[k, :k] @ y[:k]) / L[k, k]
    al = np.zeros_like(y)
    for k in range(nt - 1, -1, -1):
        al[k] = 0 if drop[k] else (y[k] - L[k+1:, k] @ al[k+1:]) / L[k, k]
    print(f"epoch {e}: nt={nt} eig min {ev[0]:.3e} max {ev[-1]:.3e} n_small {(ev < thr).sum()} dropped {drop.sum()} acc pinv32 {acc(a_pinv32):.3f} pinv64 {acc(a_pinv64):.3f} chol32 {acc(al):.3f}")
print("ridge experiments")
for e, (tr, va) in enumerate(sets):
    tr, va = tr.numpy(), va.numpy()
    Ktt = K[np.ix_(tr, tr)].astype(np.float32)
    Y = np.eye(c)[lab[tr].numpy()].astype(np.float32)
    acc = lambda a: float(((K[np.ix_(va, tr)].astype(np.float32) @ a).argmax(1) == lab[va].numpy()).mean())
    base = acc(np.linalg.pinv(Ktt) @ Y)
    out = []
    for fac in (1e-6, 1e-5, 1e-4, 1e-3, 1e-2):
        lam = np.float32(fac * Ktt.diagonal().max())
        A = Ktt + lam * np.eye(len(tr), dtype=np.float32)
        L = np.linalg.cholesky(A.astype(np.float64)).astype(np.float32)  # (fp32-rounded factor)
        y = np.linalg.solve(L.astype(np.float32), Y); al = np.linalg.solve(L.T.astype(np.float32), y)
        out.append(round(acc(al.astype(np.float32)), 3))
    print(f"epoch {e}: pinv {base:.3f} ridge(1e-6..1e-2) {out}")
