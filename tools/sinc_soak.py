#!/usr/bin/env python3
"""Soak of the split-f16 SincNet path: N calls of uvad_forward_wav on the cfg batch (256 x 5 s), every result compared bit for bit with the first
(the kernels are deterministic: persistent workgroups over fixed tile ranges, statistics combined in a fixed order)."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uvad_amd
from uvad_amd.synth import synth_pcm_device, seed_weights
ap = argparse.ArgumentParser(); ap.add_argument("--calls", type=int, default=2000); args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(1234)
m = uvad_amd.PyanNet(); m.build(); seed_weights(m, 1234, 4.0); m = m.to(dev).eval()
rt = m.runtime(dev)
wav = synth_pcm_device(256, 80000, 1000, dev)
ref = rt.forward_wav(wav, want_probs=False)[0].clone()
feats = rt.sincnet(wav).clone()
assert rt.sincnet_form() == "f16p"
wrong = 0
t0 = time.perf_counter()
for i in range(args.calls):
    lg = rt.forward_wav(wav, want_probs=False)[0]
    if i % 50 == 0:
        wrong += int(not torch.equal(lg, ref)) + int(not torch.equal(rt.sincnet(wav), feats))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"calls": args.calls, "checked": args.calls // 50, "calls_with_wrong_bits": wrong, "finite": bool(torch.isfinite(ref).all()),
                  "ms_per_call": dt / args.calls * 1e3, "frames_per_s": 256 * 293 * args.calls / dt}))
