#!/usr/bin/env python3
"""cfg-3 shaped feature launches without a graph (for rocprofv3 PMC passes):  python tools/fbank_loop.py [--lib x.so] [--n 6]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser(); ap.add_argument("--lib", default=None); ap.add_argument("--n", type=int, default=6)
args = ap.parse_args()
import uvad_amd
if args.lib:
    from uvad_amd import _lib
    _lib.LIB_PATH = os.path.abspath(args.lib)
dev = torch.device("cuda:0")
rt = uvad_amd.Fbank(uvad_amd.FbankConfig(num_filters=64, window_type="hamming"))._runtime(dev)
g = torch.Generator(device=dev); g.manual_seed(3)
pcm = 0.1 * torch.randn(4096, 16000, generator=g, device=dev)
for _ in range(args.n):
    f = rt.fbank(pcm)
torch.cuda.synchronize()
print("done", tuple(f.shape))
