#!/usr/bin/env python3
"""HBM streaming rates on this box with stock torch kernels (1 GiB buffers, far beyond the 256 MiB Infinity Cache):
write-only (fill), read+write (copy), read-only (sum).  Context for the G-write cost of the projection GEMMs."""
import json, torch
dev = torch.device("cuda:0")
n = 1 << 28   # 1 GiB of f32
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
w = t(lambda: a.fill_(1.0)); c = t(lambda: b.copy_(a)); r = t(lambda: a.sum())
print(json.dumps({"fill_1GiB_ms": w, "write_TBs": n * 4 / w / 1e9, "copy_ms": c, "copy_TBs_rw": 2 * n * 4 / c / 1e9, "sum_ms": r, "read_TBs": n * 4 / r / 1e9}))
