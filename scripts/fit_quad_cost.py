#!/usr/bin/env python3
"""Per-segment features and measured per-XCD spans of the quad-row kernel (needs `make STAMPS=1`): one JSON record per
(configuration, segment) -> gpurun_out/quad_cost_samples.json; fit offline (scripts/fit_quad_cost.py --fit <file>)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

BUCKETS = [0, 4, 8, 12, 16, 20, 24, 28, 32]  # entry-width buckets: (b[i-1], b[i]]


def features(batch):
    sp = batch.spmm
    ent = sp.keep
    order = sp.order
    rows = []
    for s in range(sp.n_segments):
        hist = np.zeros(len(BUCKETS) + 1)
        n_su = 0
        items = sp.items_host[sp.seg_ptr_host[s]:sp.seg_ptr_host[s + 1]]
        for (fj, nj, ub, ue) in items:
            w = np.concatenate([ent[order[k]][0].quad["widths"].sum(0) for k in range(fj, fj + nj)])  # per entry
            w = w[ub * 4:ue * 4]
            n_su += ue - ub
            hist[0] += (w == 0).sum()
            for i in range(1, len(BUCKETS)):
                hist[i] += ((w > BUCKETS[i - 1]) & (w <= BUCKETS[i])).sum()
            hist[len(BUCKETS)] += w.sum()
        rows.append(dict(hist=hist.tolist(), n_su=int(n_su), phases=len(items)))
    return rows


def collect():
    import torch
    from wdg_amd import sweep, synth
    from wdg_amd._lib import LIB_PATH
    lib = ctypes.CDLL(LIB_PATH)
    out = []
    for k, seeds in ((10, 5), (2, 10), (10, 10), (10, 3), (2, 5), (10, 7), (2, 7)):
        levels = synth.H_LEVELS_10 if k == 2 else synth.H_LEVELS_10_K10
        jobs = sweep.make_jobs(levels, range(seeds), k=k)
        for ph in (0, 12000, 24000):
            os.environ["WDG_QUAD_PHASE_NS"] = str(ph)
            b = sweep.SweepBatch(jobs, n_feat=500, gcn_hidden=0)
            for _ in range(5):
                b.spmm.launch()
            torch.cuda.synchronize()
            spans = np.zeros(8)
            reps = 5
            for _ in range(reps):
                b.spmm.launch()
                torch.cuda.synchronize()
                buf = np.zeros(256 * 8, np.uint64)
                assert lib.wdg_debug_q_stamps(buf.ctypes.data_as(ctypes.c_void_p), 256) == 0
                t = buf.reshape(256, 8).astype(np.float64) * 10e-3
                d = t[:, 1] - t[:, 0]
                spans += np.array([d[x::8].mean() for x in range(8)]) / reps
            feats = features(b)
            S = b.spmm.n_segments // 8
            for x in range(8):
                for s in range(S):
                    f = feats[x * S + s]
                    out.append(dict(k=k, seeds=seeds, phase_ns=ph, xcd=x, subs=S, span_us=float(spans[x]), **f))
            print(k, seeds, ph, "spans", " ".join(f"{v:.0f}" for v in spans), "phases", [feats[x * S]["phases"] for x in range(8)], flush=True)
            del b
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "quad_cost_samples.json"), "w"))


def fit(path):
    from scipy.optimize import nnls
    data = [d for d in json.load(open(path)) if d["subs"] == 1]
    A = np.array([d["hist"][:len(BUCKETS)] + [d["phases"], 1.0] for d in data])
    y = np.array([d["span_us"] for d in data])
    x, res = nnls(A, y)
    names = ["w=0"] + [f"w<={b}" for b in BUCKETS[1:]] + ["per phase", "const"]
    for n, v in zip(names, x):
        print(f"{n:10s} {v * 1000 if n.startswith('w') else v:10.3f} {'ns per entry' if n.startswith('w') else 'us'}")
    pred = A @ x
    print("rms error %.2f us, max %.2f us over %d samples" % (np.sqrt(((pred - y) ** 2).mean()), np.abs(pred - y).max(), len(y)))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--fit":
        fit(sys.argv[2])
    else:
        collect()
