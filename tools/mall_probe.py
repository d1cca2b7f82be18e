#!/usr/bin/env python3
"""Does a producer -> consumer hand-off of S MiB stay on chip (L2 4 MiB per XCD / 256 MiB Infinity Cache) on this GPU?  A buffer of S MiB is
written (fill) and then read (sum) repeatedly, plain torch kernels, for S from 8 MiB to 1 GiB; with and without 1 GiB of unrelated
streaming traffic (a copy of another buffer) between the write and the read.  Rates far above what HBM delivers at 1 GiB mean the
hand-off never went to HBM: the sizing question for a time-chunked projection -> recurrence hand-off of the gate matrix.
    python tools/mall_probe.py"""
import json, torch
dev = torch.device("cuda:0")
big_a = torch.empty(1 << 28, device=dev); big_b = torch.empty(1 << 28, device=dev)   # 1 GiB each: the polluting traffic
def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
pollute_ms = timed(lambda: big_b.copy_(big_a), 5)
out = []
for mib in (8, 16, 32, 64, 128, 192, 256, 512, 1024):
    n = mib * (1 << 20) // 4
    x = torch.empty(n, device=dev)
    reps = max(5, 2048 // mib)
    w = timed(lambda: x.fill_(1.0), reps)
    r = timed(lambda: x.sum(), reps)
    def wr(): x.fill_(2.0); x.sum()
    both = timed(wr, reps)
    def wpr(): x.fill_(3.0); big_b.copy_(big_a); x.sum()
    withp = timed(wpr, 5) - pollute_ms
    out.append({"MiB": mib, "write_TBs": round(n * 4 / w / 1e9, 2), "read_TBs": round(n * 4 / r / 1e9, 2), "write_then_read_TBs": round(2 * n * 4 / both / 1e9, 2),
                "write_then_read_TBs_with_2GiB_of_other_traffic_in_between": round(2 * n * 4 / max(withp, 1e-6) / 1e9, 2)})
    print(json.dumps(out[-1]), flush=True)
    del x
print("MALL_PROBE", json.dumps({"pollute_copy_1GiB_ms": round(pollute_ms, 3), "rows": out}))
