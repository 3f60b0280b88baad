#!/usr/bin/env python3
"""The modelled tape cut at several prices of a phase switch (WDG_QUAD_PHASE_NS), isolated launches and inside the step (dev tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "scripts"))
import torch

import bench_configs as bc
from wdg_amd import sweep as sw, synth

for n_nodes, k, seeds in ((2000, 10, 5), (4000, 10, 5), (2000, 2, 10)):
    h_levels = synth.H_LEVELS_10 if k == 2 else synth.H_LEVELS_10_K10
    jobs = sw.make_jobs(h_levels, range(seeds), k=k, n_nodes=n_nodes)
    sb = sw.SweepBatch(jobs, n_feat=500, gcn_hidden=64)
    out = []
    for price in (None, 0, 500, 1000, 2000, 4000, 8000):
        sb.spmm._set_segments(price)
        iso = bc.timed(sb.spmm.launch, 20)[1]
        step = bc.timed(sb.step, 20)[1]
        out.append(f"{price}: {iso:.1f} / {step:.1f}")
    print(f"N={n_nodes} k={k}: phase price -> launch / step us:  " + "   ".join(out), flush=True)
    del sb
    torch.cuda.empty_cache()
