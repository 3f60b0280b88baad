#!/usr/bin/env python3
"""gram_map launch on the sweep's shape: 55 feature matrices of 2000 x 500 (dev tool).  usage: time_gram.py [n_mats] [n] [f]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from wdg_amd import ops

m, n, f = (int(a) for a in (sys.argv[1:4] + ["55", "2000", "500"][len(sys.argv) - 1:]))
mats = [torch.randn(n, f, device="cuda") for _ in range(m)]
for linear, arccos in ((True, True), (True, False), (False, True)):
    gb = ops.GramBatch(mats, linear=linear, arccos=arccos)
    for _ in range(2):
        gb.launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gb.launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    outs = int(linear) + int(arccos)
    print(f"WDG_GRAM_SPLIT={os.environ.get('WDG_GRAM_SPLIT', '')} linear={linear} arccos={arccos}: {m} Grams of {n} x {f}: {ms:.3f} ms per launch = "
          f"{m * n * n * f * 2 / ms * 1e-9:.1f} TFLOP/s counted on the full square, {m * outs * n * n * 4 / ms * 1e-6:.0f} GB/s written")
    del gb
