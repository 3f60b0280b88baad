#!/usr/bin/env python3
"""Quad-row kernel builds (scripts/dev/build_quad_variants.sh) on the literal N = 4000 shard (HALF slabs) and the N = 2000 shard:
isolated launches, modelled cut, results verified against the CSR kernels (dev tool)."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VDIR = os.path.join(ROOT, "when-do-gnns-help_amd", "lib", "variants")
WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch
from wdg_amd import sweep, synth
out = {}
for nodes in (4000, 2000):
    jobs = sweep.make_jobs(synth.H_LEVELS_10_K10, range(5), k=10, n_nodes=nodes)
    batch = sweep.SweepBatch(jobs, n_feat=500, gcn_hidden=0)
    batch.spmm.verify()
    for _ in range(5):
        batch.spmm.launch()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            batch.spmm.launch()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 20 * 1e3)
    out["n%%d_us" %% nodes] = round(best, 1)
    del batch
print("RESULT " + json.dumps(out))
''' % ROOT
names = sys.argv[1:] or sorted(os.path.basename(p)[len("libwdg_hip_"):-3] for p in glob.glob(os.path.join(VDIR, "libwdg_hip_t*.so")))
for n in names:
    env = dict(os.environ, WDG_LIB_PATH=os.path.join(VDIR, f"libwdg_hip_{n}.so"))
    r = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True, timeout=900)
    res = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    print(n, res[0][7:] if res else (r.stderr or r.stdout)[-500:], flush=True)
