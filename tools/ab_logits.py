#!/usr/bin/env python3
"""Digest of the cfg-2 logits for one build of the library (A/B of kernel variants that must not change a bit):
    python tools/ab_logits.py [--lib path/to/libuvad_variant.so] [--tile 0|4|16] [--batch 256]"""
import argparse, hashlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--tile", type=int, default=0)
ap.add_argument("--batch", type=int, default=256)
args = ap.parse_args()
import uvad_amd
from uvad_amd import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from uvad_amd.synth import seed_weights, synth_pcm_device
dev = torch.device("cuda:0")
m = uvad_amd.PyanNet2(encoding_dim=64); m.build(); seed_weights(m, 1234, 4.0)
m.attach_fbank(uvad_amd.FbankConfig(num_filters=64, window_type="hamming")); m = m.to(dev).eval()
rt = m.runtime(dev)
rt.set_recurrent_tile(args.tile)
pcm = synth_pcm_device(args.batch, 160000, seed=42, device=dev)
lg, _ = rt.forward(pcm, want_probs=False)
h = hashlib.sha256(lg.cpu().numpy().tobytes()).hexdigest()[:16]
print(json.dumps({"lib": args.lib or "default", "tile": rt.recurrent_tile(), "sha256_16": h, "finite": bool(torch.isfinite(lg).all()),
                  "sum": float(lg.double().sum()), "first": lg[0, :3].tolist()}))
