"""CPU baseline leg for bench.py (ORACLE side - test/measurement infrastructure, not product code).

Runs the sweep job the way the reference does it on the host (`synthetic_plot.py:81-109` minus the
kernel-regression metric): dense `A + I`, `normalize`, DENSE `torch.spmm(adj, features)` on torch's CPU
threads (the reference's actual aggregation path on synthetic graphs), then the edge/label
metrics.  The metric arithmetic comes from the pinned oracle (oracle.py / wdg_oracle.c) so this leg reports the
same numbers as the GPU path; its timing is a reported baseline, not a target.
"""
import time

import numpy as np
import torch

from . import oracle as orc


def run_job(src, dst, labels, x, n, n_classes):
    """-> (Y [n,F] torch fp32, metrics tuple, edges)"""
    adj = torch.zeros((n, n), dtype=torch.float32)
    adj[torch.from_numpy(src), torch.from_numpy(dst)] = 1.0
    adj = adj + torch.eye(n)                                    # synthetic_plot.py:92
    rowsum = adj.sum(1)
    r_inv = 1.0 / rowsum
    r_inv[torch.isinf(r_inv)] = 0.0
    adj = r_inv[:, None] * adj                                  # utils/util_funcs.py:29-36
    y = torch.spmm(adj, x)                                      # utils/homophily_plot.py:246 (dense x dense)
    rowptr, col, _ = orc.coo_to_csr(src, dst, n, None, orc.ADD_SELF_LOOPS)
    st = orc.edge_label_stats(rowptr, col, labels, n_classes)
    metrics = (orc.edge_homophily_dense(st), orc.node_homophily_dense(st), orc.class_homophily_dense(st, labels),
               orc.adjusted_homophily_dense(st, labels), orc.label_informativeness(st, labels))
    return y, metrics, int(col.shape[0])


def time_sample(jobs, n_feat, budget_s=12.0, threads=None):
    """Time the reference-pattern CPU path on a bounded sample of the bench workload.

    jobs: list of wdg_amd.sweep.Job.  Graph / feature generation is excluded from the timed region (inputs
    resident in host memory, as they are resident in HBM for the GPU leg).  Returns edges/s and bookkeeping."""
    from wdg_amd import synth  # input generator only (numpy); no GPU code is touched
    if threads:
        torch.set_num_threads(threads)
    inputs = []
    feats = {}
    for j in jobs:
        if j.seed not in feats:
            feats[j.seed] = torch.from_numpy(synth.features(j.n_nodes, n_feat, j.seed))
        src, dst, lab = synth.regular_graph(j.n_nodes, j.n_classes, j.k, j.h, j.seed)
        inputs.append((src, dst, lab, feats[j.seed], j.n_nodes, j.n_classes))
    orc.build()
    run_job(*inputs[0])  # warm-up (thread pool, page faults)
    edges, passes, t0 = 0, 0, time.perf_counter()
    while True:
        for inp in inputs:
            _, _, e = run_job(*inp)
            edges += e
        passes += 1
        el = time.perf_counter() - t0
        if el >= budget_s or passes >= 50:
            break
    return dict(edges_per_s=edges / el, seconds=el, passes=passes, graphs=len(inputs) * passes,
                cores=torch.get_num_threads())


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def baseline_record(jobs, n_feat, budget_s=12.0):
    """The `cpu_baseline` object of bench.py's JSON line: the all-threads figure is `value`; a 1-thread figure of the same
    sample rides along (BASELINE.md section 3 asks for both).  About 3/4 of the budget goes to the all-threads run."""
    all_threads = torch.get_num_threads()
    cb = time_sample(jobs, n_feat, budget_s=budget_s * 0.75)
    one = time_sample(jobs, n_feat, budget_s=budget_s * 0.25, threads=1)
    torch.set_num_threads(all_threads)
    hs = sorted({j.h for j in jobs})
    return {"value": cb["edges_per_s"], "unit": "edges/s", "cores": cb["cores"], "kind": "port",
            "cpu_model": cpu_model(), "value_1_thread": one["edges_per_s"],
            "sample": f"{len(jobs)} graphs (seed {jobs[0].seed}, {len(hs)} h-levels, k={jobs[0].k}) x {cb['passes']} passes = "
                      f"{cb['graphs']} graph evaluations in {cb['seconds']:.1f} s on {cb['cores']} torch threads (+ {one['graphs']} "
                      f"evaluations in {one['seconds']:.1f} s on 1 thread): dense torch.spmm aggregation as the reference runs it "
                      f"(synthetic_plot.py:92, utils/homophily_plot.py:246) + the edge/label metrics computed by the C oracle "
                      f"(oracle/wdg_oracle.c), NOT by the reference's Python loops (utils/homophily_plot.py:81-186), so this "
                      f"understates the reference's cost"}
