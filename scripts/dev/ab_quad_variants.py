#!/usr/bin/env python3
"""A/B of quad-row kernel builds (scripts/dev/build_quad_variants.sh) on the bench workloads, one subprocess per library:
results checked against the CSR kernel (SpmmBatch.verify), isolated launches (best of 5 x 20) and the launch inside the
sweep step (bench.py's own event pairs).  usage: ab_quad_variants.py [variant ...]   (default: every built variant)"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VDIR = os.path.join(ROOT, "when-do-gnns-help_amd", "lib", "variants")

WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch
from wdg_amd import sweep, synth
out = {}
for k, seeds, levels, nodes in ((10, 5, synth.H_LEVELS_10_K10, 2000), (2, 10, synth.H_LEVELS_10, 2000), (10, 5, synth.H_LEVELS_10_K10, 4000)):
    jobs = sweep.make_jobs(levels, range(seeds), k=k, n_nodes=nodes)
    batch = sweep.SweepBatch(jobs, n_feat=500, gcn_hidden=0)
    if not os.environ.get("AB_NO_VERIFY"):  # (timing-only experiment builds put their results elsewhere)
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
    out["k%%d%%s_isolated_us" %% (k, "" if nodes == 2000 else "_n%%d" %% nodes)] = round(best, 1)
    del batch
print("RESULT " + json.dumps(out))
''' % ROOT


def main():
    names = sys.argv[1:] or sorted(os.path.basename(p)[len("libwdg_hip_"):-3] for p in glob.glob(os.path.join(VDIR, "libwdg_hip_*.so")))
    for n in names:
        env = dict(os.environ, WDG_LIB_PATH=os.path.join(VDIR, f"libwdg_hip_{n}.so"))
        line = {"variant": n}
        r = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True, timeout=900)
        res = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if r.returncode or not res:
            line["error"] = (r.stderr or r.stdout)[-600:]
        else:
            line.update(json.loads(res[0][7:]))
            b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "200", "--warmup", "20", "--cpu-budget", "0",
                                "--secondary", "0", "--full-metrics", "0", "--cold", "0", "--configs", "0", "--train", "0", "--projection", "0", "--whole", "0"], env=env, capture_output=True, text=True, timeout=900)
            try:
                j = json.loads([l for l in b.stdout.splitlines() if l.startswith("{")][-1])
                line.update(step_ms=j["ms_per_step"], launch_us=j["roofline"].get("avg_launch_us"), frac=j["roofline"]["frac"])
            except Exception as e:  # noqa: BLE001
                line["bench_error"] = f"{e}: {(b.stderr or b.stdout)[-400:]}"
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
