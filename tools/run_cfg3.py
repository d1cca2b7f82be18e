#!/usr/bin/env python3
"""BASELINE cfg 3: 4096 streams x 1 s chunks (100 frames per stream per step), FEATURE KERNEL ONLY, the
per-chunk loop of 100 steps captured into one hipGraph (torch.cuda.CUDAGraph capturing the uvad_fbank launches:
the C ABI does no allocation / sync, so it is capturable).  Reports frames/s and the algorithmic HBM rate
(896 B per frame at 64 mels, f32 PCM) against the 8 TB/s peak."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uvad_amd

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=4096)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--int16", action="store_true")
ap.add_argument("--chunk", type=int, default=16000, help="samples per stream and step (cfg 3: 1 s)")
ap.add_argument("--lib", default=None, help="another build of libuvad.so (A/B on one box)")
args = ap.parse_args()
if args.lib:
    from uvad_amd import _lib
    _lib.LIB_PATH = os.path.abspath(args.lib)
dev = torch.device("cuda:0")
B, C, F = args.streams, args.chunk, 64
rt = uvad_amd.Fbank(uvad_amd.FbankConfig(num_filters=F, window_type="hamming"))._runtime(dev)
g = torch.Generator(device=dev); g.manual_seed(3)
pcm = [0.1 * torch.randn(B, C, generator=g, device=dev) for _ in range(4)]      # 4 distinct chunks, cycled
if args.int16:
    pcm = [(p * 32767).to(torch.int16) for p in pcm]
T = rt.num_frames(C)
feats = torch.empty(B, T, F, device=dev)
import ctypes as C_
fn = rt.lib.uvad_fbank_i16 if args.int16 else rt.lib.uvad_fbank
side = torch.cuda.Stream(device=dev)
with torch.cuda.stream(side):
    for p in pcm:                                                                   # warm-up outside capture
        rt._check(fn(rt.ctx, p.data_ptr(), B, C, feats.data_ptr(), C_.c_void_p(side.cuda_stream)))
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for i in range(args.steps):
            rt._check(fn(rt.ctx, pcm[i % 4].data_ptr(), B, C, feats.data_ptr(), C_.c_void_p(side.cuda_stream)))
graph.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.reps):
    graph.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.reps
frames = B * T * args.steps
bpf = (320 if args.int16 else 640) + 4 * F
print(json.dumps({"config": f"{B} streams x {C / 16000:g} s chunks x {args.steps} steps, hipGraph, fbank only, pcm {'int16' if args.int16 else 'f32'}",
                  "frames_per_s": frames / dt, "ms_per_graph": dt * 1e3, "us_per_step": dt / args.steps * 1e6,
                  "algorithmic_GBs": frames * bpf / dt / 1e9, "frac_of_8TBs": frames * bpf / dt / 8e12}))
