#!/usr/bin/env python3
"""HBM write / copy bandwidth probes (context for the group_points roofline)."""
import torch, time
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for mb in (256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
    ms = t(lambda: x.fill_(1.0)); print("fill   %5d MB: %.3f ms  %.0f GB/s (write)" % (mb, ms, n * 4 / ms / 1e6))
    ms = t(lambda: y.copy_(x)); print("copy   %5d MB: %.3f ms  %.0f GB/s (read+write)" % (mb, ms, 2 * n * 4 / ms / 1e6))
    ms = t(lambda: x.sum()); print("sum    %5d MB: %.3f ms  %.0f GB/s (read)" % (mb, ms, n * 4 / ms / 1e6))
