#!/usr/bin/env python3
"""Latency experiment: one BASELINE cfg-2 batch submitted as 1 / 2 / 4 concurrent parts (ForwardPipeline slots)."""
import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import uvad_amd
from uvad_amd.synth import seed_weights, synth_pcm_device
dev = torch.device("cuda:0")
m = uvad_amd.PyanNet2(encoding_dim=64); m.build(); seed_weights(m, 1234, 4.0)
m.attach_fbank(uvad_amd.FbankConfig(num_filters=64, window_type="hamming")); m = m.to(dev).eval()
pcm = synth_pcm_device(256, 160000, 42, dev)
for parts in (1, 2, 4):
    pipe = uvad_amd.ForwardPipeline(m, dev, depth=parts)
    chunks = list(pcm.chunk(parts))
    pipe._calibrate(chunks[0])
    for _ in range(2):
        [p.wait() for p in [pipe.submit(c) for c in chunks]]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        pend = [pipe.submit(c) for c in chunks]
        [p.wait() for p in pend]
    dt = (time.perf_counter() - t0) / 10
    print(f"one 256 x 10 s batch as {parts} concurrent part(s): {dt*1e3:.2f} ms per batch = {256000/dt/1e6:.1f} M frames/s", flush=True)
    if parts == 2:   # the same two halves with the second one submitted a little later (host spin), so that they run out of phase
        for delay_us in (150, 300, 600, 1000):
            def run():
                a = pipe.submit(chunks[0])
                t = time.perf_counter()
                while (time.perf_counter() - t) * 1e6 < delay_us:
                    pass
                b = pipe.submit(chunks[1])
                a.wait(); b.wait()
            run(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                run()
            dt = (time.perf_counter() - t0) / 10
            print(f"   two halves, second submitted {delay_us} us later: {dt*1e3:.2f} ms per batch", flush=True)
    pipe.close()
