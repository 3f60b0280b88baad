#!/usr/bin/env python3
"""The batched kernel-regression solver on the C3 shard's 20 000 problems (dev tool): launch time, and - A/B in two processes,
WDG_KR_KERNEL=rank1 is round 2's solver - the per-problem hit counts written to gpurun_out/ for comparison.
    python scripts/dev/time_kr_batch.py [seeds] [epochs] [out.npy]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from wdg_amd import sweep, synth

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 100
jobs = sweep.make_jobs(synth.H_LEVELS_10_K10, range(seeds), k=10)
sb = sweep.SweepBatch(jobs, n_feat=500, gcn_hidden=0)
sb.prepare_full(epochs=epochs, sample_max=500)
sb.step()
sb.kr_sets.launch()
sb.gram.launch()
sb.ge.launch()
torch.cuda.synchronize()
for _ in range(2):
    sb.kr.launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    sb.kr.launch()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
n = sb.kr.n_jobs
flops = n * (300 ** 3 / 3 + 2 * 300 * 300 * 8 + 2 * 200 * 300 * 8)
print(f"kernel={os.environ.get('WDG_KR_KERNEL', 'blocked')}: {n} regressions in {ms:.2f} ms = {ms * 1e3 / n * 256:.1f} us per problem and CU, "
      f"{flops / ms * 1e-9:.1f} TFLOP/s; ridged {int(sb.kr.ridged().sum())}", flush=True)
correct = sb.kr.correct[:n].cpu().numpy()
print("hits: mean %.2f min %d max %d" % (correct.mean(), correct.min(), correct.max()))
if len(sys.argv) > 3:
    np.save(sys.argv[3], correct)
