#!/usr/bin/env python3
"""BASELINE cfg 5: 512 concurrent real-time feeds, 20 ms chunks (320 samples -> 2 frames per step), one GPU.
Reports p50/p99 wall latency per step (host submit -> logits visible on the host-synchronised stream) and the
real-time factor (compute time / audio time).  Causal model (lstm.bidirectional=False), carried (h, c)."""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uvad_amd
from uvad_amd.synth import seed_weights

ap = argparse.ArgumentParser()
ap.add_argument("--feeds", type=int, default=512)
ap.add_argument("--chunk", type=int, default=320)
ap.add_argument("--seconds", type=float, default=60.0)
ap.add_argument("--lib", default=None, help="another build of libuvad.so (A/B on one box)")
args = ap.parse_args()
if args.lib:
    from uvad_amd import _lib
    _lib.LIB_PATH = os.path.abspath(args.lib)
dev = torch.device("cuda:0")
m = uvad_amd.PyanNet2(lstm={"bidirectional": False}, encoding_dim=64); m.build(); seed_weights(m, 1234, 4.0)
m.attach_fbank(uvad_amd.FbankConfig(num_filters=64, window_type="hamming")); m = m.to(dev).eval()
rt = m.runtime(dev)
B, C = args.feeds, args.chunk
steps = int(args.seconds * 16000 / C)
g = torch.Generator(device=dev); g.manual_seed(5)
audio = 0.1 * torch.randn(B, 64 * C, generator=g, device=dev)        # 64 distinct chunks, cycled
st = rt.stream_open(B, C)
lat, dev_ms = [], []
frames = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(20):                                                    # warm-up
    rt.stream_step(st, audio[:, (i % 64) * C:(i % 64 + 1) * C].contiguous())
torch.cuda.synchronize()
st = rt.stream_open(B, C)
t_all = time.perf_counter()
for i in range(steps):
    x = audio[:, (i % 64) * C:(i % 64 + 1) * C].contiguous()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = rt.stream_step(st, x)
    torch.cuda.synchronize()
    lat.append(time.perf_counter() - t0)
    frames += out.shape[1]
wall = time.perf_counter() - t_all
lat = np.array(lat) * 1e3
# a second, shorter pass with HIP events around the step (kept out of the latency loop: two event records cost the host ~3 us)
for i in range(min(steps, 300)):
    x = audio[:, (i % 64) * C:(i % 64 + 1) * C].contiguous()
    torch.cuda.synchronize()
    e0.record()
    rt.stream_step(st, x)
    e1.record()
    torch.cuda.synchronize()
    dev_ms.append(e0.elapsed_time(e1))      # device time between the two event records: the step's kernel plus its launch gap
audio_s = steps * C / 16000.0
print(json.dumps({"config": f"{B} feeds x {C}-sample chunks, {audio_s:.0f} s of audio per feed", "steps": steps,
                  "frames_per_feed": frames, "p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)),
                  "max_ms": float(lat.max()), "rtf": float(lat.sum() / 1e3 / audio_s),
                  "device_ms_p50": float(np.percentile(dev_ms, 50)),
                  "host_launch_and_sync_ms_p50": float(np.percentile(lat, 50) - np.percentile(dev_ms, 50)),
                  "what": "p50_ms = host wall time of one step (submit -> synchronised); device_ms = HIP events around the step's one launch (kernel + launch "
                          "gap on the device); the difference is the host's launch and synchronisation cost",
                  "aggregate_frames_per_s": B * frames / float(lat.sum() / 1e3)}))
