#!/usr/bin/env python3
"""Per-stage device times of one cfg-2 step (B = 256 x 10 s) submitted alone on one stream, for one build of the library.
    python tools/stage_times.py [--lib path/to/libuvad_variant.so] [--batch 256] [--reps 10]
Used to A/B kernel variants (e.g. `make -C .../csrc F16P_BK=32` builds) on one box in one gpurun call."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--mode", default="f16p", choices=["f16p", "f16p_stream", "f32", "f16p3"])
ap.add_argument("--tile", type=int, default=0, help="recurrent form: 0 = by estimated time, 4, 16")
ap.add_argument("--seconds", type=float, default=10.0)
ap.add_argument("--mels", type=int, default=64)
ap.add_argument("--window", default="hamming")
ap.add_argument("--chunks", type=int, default=0, help="time chunks: 0 = automatic, 1 = off")
args = ap.parse_args()
import uvad_amd
from uvad_amd import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from uvad_amd.synth import seed_weights, synth_pcm_device
dev = torch.device("cuda:0")
m = uvad_amd.PyanNet2(encoding_dim=args.mels); m.build(); seed_weights(m, 1234, 4.0)
m.attach_fbank(uvad_amd.FbankConfig(num_filters=args.mels, window_type=args.window)); m = m.to(dev).eval()
rt = m.runtime(dev)
rt.set_gemm_mode(args.mode)
rt.set_recurrent_tile(args.tile)
rt.set_time_chunks(args.chunks)
pcm = synth_pcm_device(args.batch, int(args.seconds * 16000), seed=42, device=dev)
for _ in range(3):
    rt.forward(pcm, want_probs=False)
torch.cuda.synchronize()
rt.set_timing(True)
acc = {}
for _ in range(args.reps):
    rt.forward(pcm, want_probs=False)
    for k, v in rt.timing_ms().items():
        acc[k] = acc.get(k, 0.0) + v / args.reps
print(json.dumps({"lib": args.lib or "default", "mode": args.mode, "batch": args.batch, "tile": rt.recurrent_tile(), "ms": {k: round(v, 4) for k, v in acc.items()}}))
print(json.dumps({"per_layer_ms (projection, recurrence)": [[round(a, 4), round(b, 4)] for a, b in rt.layer_timing_ms()]}))
