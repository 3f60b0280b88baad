#!/usr/bin/env python3
"""Error of the fused transform's first product against fp64: k-ordered fp32 chain vs split bf16 operands (dev tool; the data of
tests/test_gpu_kernels.py::test_mlp2_split_operands_are_as_accurate_as_the_fp32_chain and two plainer sets)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from wdg_amd import ops

rng = np.random.default_rng(77)
m, k, h = 4096, 500, 8


def run(a, w0, tag):
    ref = a.astype(np.float64) @ w0.astype(np.float64)
    mag = np.abs(a).astype(np.float64) @ np.abs(w0).astype(np.float64)
    eye = np.eye(h, dtype=np.float32)
    for split in ("0", "1"):
        os.environ["WDG_MLP2_SPLIT"] = split
        z = torch.full((m, h), float("nan"), device="cuda")
        b = ops.Mlp2Batch([(torch.from_numpy(a).cuda(), torch.from_numpy(w0).cuda(), None, torch.from_numpy(eye).cuda(), None, z)], relu=False)
        b.launch()
        torch.cuda.synchronize()
        e = np.abs(z.cpu().numpy().astype(np.float64) - ref)
        print(f"{tag:28s} split={split}: max |err| / sum|x||w| {np.max(e / mag):.3e}  mean {np.mean(e / mag):.3e}   max |err| / max|out| {e.max() / np.abs(ref).max():.3e}")


a = (rng.standard_normal((m, k)) * 10.0 ** rng.uniform(-6, 2, (m, k))).astype(np.float32)
w0 = (rng.standard_normal((k, h)) * 10.0 ** rng.uniform(-3, 1, (k, h))).astype(np.float32)
run(a, w0, "eight decades per row")
run(rng.standard_normal((m, k)).astype(np.float32), (rng.standard_normal((k, h)) / np.sqrt(k)).astype(np.float32), "standard normal")
run(np.abs(rng.standard_normal((m, k))).astype(np.float32) / k, np.abs(rng.standard_normal((k, h))).astype(np.float32), "positive (aggregated rows)")
