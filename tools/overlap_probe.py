#!/usr/bin/env python3
"""Diagnostic: do two uvad_forward steps submitted to two HIP streams overlap?  Tries several stream-creation orders
(HIP streams share a small pool of hardware queues; two streams on one queue serialise)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uvad_amd
from uvad_amd.runtime import VadRuntime
from uvad_amd.synth import seed_weights, synth_pcm_device

dev = torch.device("cuda:0")
model = uvad_amd.PyanNet2(encoding_dim=64)
model.build()
seed_weights(model, 1234, 4.0)
model.attach_fbank(uvad_amd.FbankConfig(num_filters=64, window_type="hamming"))
model = model.to(dev).eval()
cfg = {"encoding_dim": 64, "lstm": model.hparams.lstm, "linear": model.hparams.linear}
pcm = synth_pcm_device(256, 160000, 42, dev)


def mk():
    r = VadRuntime(device=dev, fbank=model._fbank_cfg, model=cfg)
    r.load_state_dict(model.state_dict())
    return r


def run(rts, streams, steps=10, label=""):
    for i in range(len(rts)):
        with torch.cuda.stream(streams[i]):
            rts[i].forward(pcm, want_probs=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        i = k % len(rts)
        with torch.cuda.stream(streams[i]):
            rts[i].forward(pcm, want_probs=False)
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step", flush=True)


# order 1: ctx, stream, ctx, stream
a = mk(); sa = torch.cuda.Stream(device=dev); b = mk(); sb = torch.cuda.Stream(device=dev)
run([a, b], [sa, sb], label="ctx,stream,ctx,stream")
run([a], [sa], label="  one ctx alone")
# order 2: streams first
s1 = torch.cuda.Stream(device=dev); s2 = torch.cuda.Stream(device=dev)
run([a, b], [s1, s2], label="two streams created back to back")
# order 3: more streams: try all pairs of 6
ss = [torch.cuda.Stream(device=dev) for _ in range(6)]
for i in range(6):
    for j in range(i + 1, 6):
        run([a, b], [ss[i], ss[j]], steps=6, label=f"pool pair ({i},{j})")
# high priority streams
h1 = torch.cuda.Stream(device=dev, priority=-1); h2 = torch.cuda.Stream(device=dev, priority=0)
run([a, b], [h1, h2], label="priority -1 / 0")
