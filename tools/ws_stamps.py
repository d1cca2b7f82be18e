#!/usr/bin/env python3
"""Diagnostic (stamped variant builds of gemm_f16p_ws.hip only): per-phase cycle shares of a k-block.  python tools/ws_stamps.py --lib X.so"""
import os, sys, argparse
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser(); ap.add_argument("--lib", required=True); args = ap.parse_args()
import uvad_amd
from uvad_amd import _lib
_lib.LIB_PATH = os.path.abspath(args.lib)
from uvad_amd.synth import seed_weights
dev = torch.device("cuda:0")
m = uvad_amd.PyanNet2(encoding_dim=64); m.build(); seed_weights(m, 1234, 4.0); m = m.to(dev).eval()
rt = m.runtime(dev); rt.set_recurrent_tile(16)
g = torch.Generator(device=dev); g.manual_seed(11)
feats = torch.randn(256, 1000, 64, generator=g, device=dev) * 4.0 - 8.0
for _ in range(3): rt.classify(feats, want_probs=False)
torch.cuda.synchronize()
c = rt._ws[-8192:].view(torch.int32).view(64, 32).cpu().double()
names = ["MFMA body (stamp5 -> next stamp0)", "vmcnt+lgkm wait", "barrier", "MFMA pair 0", "gap 0 (stores + DMA hi)", "pair 1 + gap 1 (DMA lo + 2 reads)"]
for w, off in (("wave 0", 1), ("wave 3", 9)):
    v = c[:, off:off + 6] * 16.0
    tot = v.sum(1)
    tiles = c[:, 0]
    print(w, "cycles per k-block (mean over 64 workgroups; last launch = layer 3, K = 256):")
    kb = (tiles.mean() - 2) / 4 * 16 if False else None
    for i, n in enumerate(names):
        print(f"   {n:42s} {v[:, i].mean() / (62.5 * 16):8.1f}")
    print(f"   total {tot.mean() / (62.5 * 16):8.1f}")
