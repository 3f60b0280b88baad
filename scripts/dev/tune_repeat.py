import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch
from wdg_amd import sweep, synth
jobs = sweep.make_jobs(synth.H_LEVELS_10_K10, range(5), k=10, n_nodes=2000)
for rep in range(6):
    sb = sweep.SweepBatch(jobs, n_feat=500)
    sb.step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    best = sb.tune()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # measure launches
    ts = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); sb.spmm.launch(); b.record(); sb.step_rest(); ts.append((a, b))
    torch.cuda.synchronize()
    us = sorted(a.elapsed_time(b) for a, b in ts)
    print(f"rep {rep}: tune {dt*1e3:.0f} ms, launch median {us[15]*1e3:.1f} us, burst {({k: round(v*1e3,1) for k,v in best[3].items()}) if len(best)>3 else None}", flush=True)
    del sb
