#!/usr/bin/env python3
"""What this device's write path does with plain streaming stores (torch fill / copy): the yardstick for the aggregation's Y stores"""
import torch
for mb in (205, 410, 1024):
    n = mb * 1024 * 1024 // 4
    y = torch.empty(n, device="cuda"); x = torch.ones(n, device="cuda")
    for name, fn in (("fill", lambda: y.fill_(2.0)), ("copy", lambda: y.copy_(x))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 10 * 1e3
        print(f"{name} {mb:5d} MB: {us:7.1f} us = {mb * 1.048576 / us * 1e3 / 1e3:5.2f} TB/s written" + (" (+ as much read)" if name == "copy" else ""), flush=True)
