#!/usr/bin/env python3
"""Three answers for the SAME regressions of the golden fixtures (tests/golden/kr_epochs.npz: the reference's node sets and the
accuracies it computed in each epoch): (a) the device solver (Cholesky, ridge on flagged blocks), (b) np.linalg.pinv on the host
applied to the blocks gathered from the DEVICE kernels (what ridge='pinv' / the API twin's re-solve do), (c) the reference's own
recorded accuracy.  Which of (a), (b) is closer to (c) on the blocks the solver flags?  (dev tool; run on the GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from _golden import KR_FIXTURES, load_kr
from test_gpu_kr_epochs import _device_inputs
from wdg_amd import ops

try:
    from threadpoolctl import threadpool_info, threadpool_limits
    print("threadpools:", [(d.get("user_api"), d.get("internal_api"), d.get("num_threads")) for d in threadpool_info()])
except ImportError:
    threadpool_limits = None
print("rows (of 196 - 200 validation rows; texas 73) by which an epoch's accuracy differs from what the REFERENCE recorded in that epoch")
print(f"{'fixture':18s} {'clf':11s} {'kernel':8s} {'flagged':>8s} | flagged blocks, max / mean: {'device ridge':>14s} {'pinv(device K)':>15s} {'pinv(host GEMM K)':>18s} | other blocks, max: {'device':>7s} {'pinv(dev K)':>12s}")
t_pinv, n_pinv = 0.0, 0
for name in KR_FIXTURES:
    kr = load_kr(name)
    h, x, lab = _device_inputs(name)
    gb = ops.GramBatch([h, x])
    gb.launch()
    lab_h = lab.cpu().numpy().astype(np.int64)
    c = int(lab_h.max()) + 1
    eye = np.eye(c, dtype=np.float32)
    for clf, kern in (("kernel_reg0", gb.k_linear), ("kernel_reg1", gb.k_arccos)):
        rec = kr[clf]
        problems = []
        for tr, va in rec["node_sets"]:
            trd, vad = torch.from_numpy(tr).cuda().to(torch.int32), torch.from_numpy(va).cuda().to(torch.int32)
            problems += [(kern[0], trd, vad, lab), (kern[1], trd, vad, lab)]
        kb = ops.KrBatch(problems, c)
        kb.launch()
        torch.cuda.synchronize()
        acc = kb.accuracy().cpu().numpy().reshape(-1, 2).astype(np.float64)
        ridged = kb.ridged().cpu().numpy().reshape(-1, 2)
        for which, wname, ref in ((0, "graph", rec["g_results"]), (1, "features", rec["x_results"])):
            host, host2 = np.zeros(len(rec["node_sets"])), np.zeros(len(rec["node_sets"]))
            feat = (h if which == 0 else x)
            n_layers = 0 if clf == "kernel_reg0" else 1
            for e, (tr, va) in enumerate(rec["node_sets"]):
                # (d) the reference's own per-epoch computation: sampled Gram by the host's fp32 GEMM + map + pinv
                smp = np.concatenate([tr, va])
                hs = feat[torch.from_numpy(smp).cuda().long()].cpu().numpy()
                g_ = hs @ hs.T
                if n_layers == 1:
                    d_ = np.sqrt(np.diag(g_)); nrm = d_[:, None] * d_[None, :]
                    nrm = np.where(nrm > 1e-8, nrm, 1e-8).astype(np.float32)
                    with np.errstate(invalid="ignore", divide="ignore"):
                        ac, sq = np.arccos(g_ / nrm), np.sqrt(np.square(nrm) - np.square(g_))
                    ac[np.isnan(ac)], sq[np.isnan(sq)] = 0, 0
                    g_ = (np.float32(1 / np.pi) * (g_ * (np.float32(np.pi) - ac) + sq)).astype(np.float32)
                g_ = g_ / np.float32(2)
                nt_ = len(tr)
                pred2 = g_[nt_:, :nt_] @ (np.linalg.pinv(g_[:nt_, :nt_]) @ eye[lab_h[tr]])
                host2[e] = np.mean(pred2.argmax(1) == lab_h[va])
                k = kern[which]
                trl, val_ = torch.from_numpy(tr).cuda().long(), torch.from_numpy(va).cuda().long()
                k_tt, k_vt = k[trl][:, trl].cpu().numpy(), k[val_][:, trl].cpu().numpy()
                t0 = time.perf_counter()
                pred = k_vt @ (np.linalg.pinv(k_tt) @ eye[lab_h[tr]])
                t_pinv += time.perf_counter() - t0
                n_pinv += 1
                host[e] = np.mean(pred.argmax(1) == lab_h[va])
            n_val = np.array([len(v) for _, v in rec["node_sets"]], np.float64)
            d_dev, d_host, d_host2 = np.abs(acc[:, which] - ref) * n_val, np.abs(host - ref) * n_val, np.abs(host2 - ref) * n_val
            f = ridged[:, which].astype(bool)
            fmt = lambda v: f"{v.max():5.1f} / {v.mean():5.2f}" if v.size else "    - /     -"  # noqa: E731
            print(f"{name:18s} {clf:11s} {wname:8s} {int(f.sum()):3d} / {f.size:<3d} | {'':27s} {fmt(d_dev[f]):>14s} {fmt(d_host[f]):>15s} {fmt(d_host2[f]):>18s} | {'':18s} "
                  f"{(d_dev[~f].max() if (~f).any() else 0):7.1f} {(d_host[~f].max() if (~f).any() else 0):12.1f}", flush=True)
print(f"np.linalg.pinv + products on this host: {t_pinv / max(n_pinv, 1) * 1e3:.1f} ms per block (serial, default BLAS threads)")
if threadpool_limits is not None:
    rng = np.random.default_rng(0)
    mats = [(lambda hh: (hh @ hh.T).astype(np.float32))(rng.standard_normal((300, 280)).astype(np.float32)) for _ in range(64)]
    from concurrent.futures import ThreadPoolExecutor
    for lim in (None, 1):
        ctx = threadpool_limits(limits=lim) if lim else None
        for workers in (1, 16, 64):
            t0 = time.perf_counter()
            with ThreadPoolExecutor(workers) as pool:
                list(pool.map(np.linalg.pinv, mats))
            print(f"64 pinvs of 300 x 300 fp32: BLAS limit {lim}, {workers:2d} python threads: {time.perf_counter() - t0:.3f} s", flush=True)
        if ctx is not None:
            ctx.restore_original_limits()
