#!/usr/bin/env python3
"""Time-chunked layers (uvad_set_time_chunks) on one cfg-2 batch alone on the GPU: outputs against the unchunked call (must be
bit-identical) and the step time by chunk count.
    python tools/chunk_check.py [--batch 256] [--seconds 10] [--chunks 1,0,2,4,8,16]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--seconds", type=float, default=10.0)
ap.add_argument("--chunks", default="1,0,2,4,8,16")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--mode", default="f16p")
args = ap.parse_args()
import uvad_amd
from uvad_amd.synth import seed_weights, synth_pcm_device
dev = torch.device("cuda:0")
m = uvad_amd.PyanNet2(encoding_dim=64); m.build(); seed_weights(m, 1234, 4.0)
m.attach_fbank(uvad_amd.FbankConfig(num_filters=64, window_type="hamming")); m = m.to(dev).eval()
rt = m.runtime(dev)
rt.set_gemm_mode(args.mode)
pcm = synth_pcm_device(args.batch, int(args.seconds * 16000), seed=42, device=dev)
ref = None
for n in [int(x) for x in args.chunks.split(",")]:
    rt.set_time_chunks(n)
    for _ in range(2):
        out = rt.forward(pcm, want_probs=False)[0]
    torch.cuda.synchronize()
    used = rt.time_chunks()
    if ref is None:
        ref = out.clone()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        rt.forward(pcm, want_probs=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.reps * 1e3
    rt.set_timing(True)
    rt.forward(pcm, want_probs=False)
    tm = rt.timing_ms()
    rt.set_timing(False)
    print(json.dumps({"requested": n, "chunks_used": used, "tile": rt.recurrent_tile(), "identical_to_unchunked": bool(torch.equal(out, ref)),
                      "max_diff": float((out - ref).abs().max()), "ms_per_step": round(ms, 4),
                      "stage_ms": {k: round(v, 3) for k, v in tm.items()}}), flush=True)
