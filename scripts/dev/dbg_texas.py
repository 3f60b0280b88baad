import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from _golden import load_kr
from test_gpu_kr_epochs import _device_inputs
from wdg_amd import ops
name = sys.argv[1] if len(sys.argv) > 1 else "real_texas"
kr = load_kr(name)
h, x, lab = _device_inputs(name)
gb = ops.GramBatch([h, x]); gb.launch(); torch.cuda.synchronize()
labh = lab.cpu().numpy()
c = int(labh.max()) + 1
for clf, kern in (("kernel_reg0", gb.k_linear), ("kernel_reg1", gb.k_arccos)):
    rec = kr[clf]
    K = kern[0]
    rep = gb.rep[0].cpu().numpy()
    print(clf, "nodes", len(rep), "non-identity reps", int((rep != np.arange(len(rep))).sum()), "zero diag", int((torch.diagonal(K) == 0).sum()))
    k32 = K.cpu().numpy(); k64 = k32.astype(np.float64)
    hh = h.cpu().numpy()
    for e, (tr, va) in enumerate(rec["node_sets"]):
        res = {}
        for tag, r in (("plain", None), ("deflated", gb.rep[0])):
            kb = ops.KrBatch([(K, torch.from_numpy(tr).cuda().int(), torch.from_numpy(va).cuda().int(), lab, r)], c)
            kb.launch(); torch.cuda.synchronize()
            res[tag] = (int(kb.correct[0]), int(kb.flags[0]))
        eye = np.eye(c, dtype=np.float32)
        p32 = k32[np.ix_(va, tr)] @ (np.linalg.pinv(k32[np.ix_(tr, tr)]) @ eye[labh[tr]])
        res["pinv32(devK)"] = int((p32.argmax(1) == labh[va]).sum())
        g = (hh[np.concatenate([tr, va])] @ hh[np.concatenate([tr, va])].T).astype(np.float32)
        if clf == "kernel_reg1":
            d = np.sqrt(np.diag(g)); nrm = d[:, None] * d[None, :]; nrm = np.where(nrm > 1e-8, nrm, 1e-8).astype(np.float32)
            with np.errstate(invalid="ignore"):
                ac, sq = np.arccos(g / nrm), np.sqrt(np.square(nrm) - np.square(g))
            ac[np.isnan(ac)] = 0; sq[np.isnan(sq)] = 0
            g = (np.float32(1) / np.float32(np.pi) * (g * (np.float32(np.pi) - ac) + sq)).astype(np.float32)
        g = g / np.float32(2); nt = len(tr)
        ph = g[nt:, :nt] @ (np.linalg.pinv(g[:nt, :nt]) @ eye[labh[tr]])
        res["pinv32(hostK)"] = int((ph.argmax(1) == labh[va]).sum())
        kc = k64[np.ix_(rep, rep)]
        p64 = kc[np.ix_(va, tr)] @ (np.linalg.pinv(kc[np.ix_(tr, tr)], rcond=1e-13) @ np.eye(c)[labh[tr]])
        res["pinv64 exact"] = int((p64.argmax(1) == labh[va]).sum())
        s64 = np.linalg.svd(kc[np.ix_(tr, tr)], compute_uv=False)
        print(" epoch", e, "reference", int(round(rec["g_results"][e] * len(va))), res, "rank64", int((s64 > 1e-12 * s64[0]).sum()), "of", nt,
              "uniq", len(set(rep[tr].tolist())))
