import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from wdg_amd import ops, sweep, synth
jobs = sweep.make_jobs(synth.H_LEVELS_10_K10, range(5), k=10)
feats, inputs = {}, []
for j in jobs:
    src, dst, lab = synth.regular_graph(j.n_nodes, j.n_classes, j.k, j.h, j.seed)
    feats.setdefault(j.seed, synth.features(j.n_nodes, 500, j.seed))
    inputs.append((src, dst, lab, feats[j.seed]))
coos = [(i[0], i[1], 2000) for i in inputs]
for rep in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gb = ops.GraphBatch(coos, ops.COO_ADD_SELF_LOOPS, quad=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    sb = sweep.SweepBatch(jobs, n_feat=500, gcn_hidden=0, inputs=inputs)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"rep {rep}: GraphBatch {1e3*(t1-t0):.2f} ms  SweepBatch {1e3*(t2-t1):.2f} ms", flush=True)
    del gb, sb
os.environ["WDG_SWEEP_HOST_PACK"] = "0"
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gb = ops.GraphBatch(coos, ops.COO_ADD_SELF_LOOPS, quad=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"no host pack rep {rep}: GraphBatch {1e3*(t1-t0):.2f} ms", flush=True)
os.environ["WDG_SWEEP_HOST_PACK"] = "1"
import cProfile, pstats
for rep in range(6):
    pr = cProfile.Profile()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pr.enable()
    sb = sweep.SweepBatch(jobs, n_feat=500, gcn_hidden=0, inputs=inputs)
    torch.cuda.synchronize()
    pr.disable()
    dt = time.perf_counter() - t0
    print(f"profiled rep {rep}: {dt * 1e3:.2f} ms")
    if dt > 0.03:
        st = pstats.Stats(pr)
        rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:6]
        for (fn, line, name), (cc, nc, tt, ct, callers) in rows:
            print(f"   {nc:6d} calls own {tt * 1e3:8.2f} ms  {os.path.basename(fn)}:{line} {name}")
            if name.startswith("<method 'data_ptr'") or "synchronize" in name:
                for (cfn, cline, cname), (_a, cnc, ctt, _c) in sorted(callers.items(), key=lambda kv: -kv[1][2])[:4]:
                    print(f"          from {os.path.basename(cfn)}:{cline} {cname}: {cnc} calls {ctt * 1e3:.2f} ms")
    del sb
