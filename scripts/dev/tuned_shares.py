import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import numpy as np, torch
from wdg_amd import sweep as sw, synth
np.set_printoptions(precision=3, suppress=True, linewidth=200)
for n_nodes, k, seeds in ((2000, 10, 5), (4000, 10, 5), (2000, 2, 10)):
    h_levels = synth.H_LEVELS_10 if k == 2 else synth.H_LEVELS_10_K10
    jobs = sw.make_jobs(h_levels, range(seeds), k=k, n_nodes=n_nodes)
    sb = sw.SweepBatch(jobs, n_feat=500, gcn_hidden=64)
    clock = sb.spmm.new_clock()
    for _ in range(3): sb.step()
    spans = np.zeros(8)
    for _ in range(5):
        sb.spmm.launch(clock=clock); sb.step_rest(); torch.cuda.synchronize(); spans += sb.spmm.segment_spans(clock) / 5
    print(n_nodes, k, "modelled cut spans us", spans)
    best = sb.tune()
    print("   tuned: t", best[0], "shares", best[1], "spans", best[2], best[3] if len(best) > 3 else "")
    del sb; torch.cuda.empty_cache()
