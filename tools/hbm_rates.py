#!/usr/bin/env python3
"""Streaming rates of this MI355X for the three access mixes the hot kernels have (torch elementwise kernels, 6 GB
operands, HIP events): pure read (sum), pure write (fill), copy (read + write), in-place scale (read + write of the same
lines).  The evaluator is a write stream (208 of its 232 bytes per residual block), the S x kernels are read streams."""
import json
import torch
n = 750_000_000  # doubles: 6 GB
a = torch.empty(n, dtype=torch.float64, device="cuda")
b = torch.empty(n, dtype=torch.float64, device="cuda")
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
out = {}
ms = timed(lambda: a.fill_(1.0)); out["write_only_TBps"] = 8 * n / ms / 1e9
ms = timed(lambda: a.sum()); out["read_only_TBps"] = 8 * n / ms / 1e9
ms = timed(lambda: b.copy_(a)); out["copy_TBps_read_plus_write"] = 16 * n / ms / 1e9
ms = timed(lambda: a.mul_(1.0000001)); out["inplace_scale_TBps_read_plus_write"] = 16 * n / ms / 1e9
print(json.dumps(out))
