#!/usr/bin/env python3
"""First contact of a new projection-kernel build with the GPU: small shapes first, each compared bit for bit with the tile-streaming
kernel (run under a short `timeout`: a persistent kernel that does not drain must not hold the box).  python tools/ws_check.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uvad_amd
from uvad_amd.synth import seed_weights
dev = torch.device("cuda:0")
print("start", flush=True)
for F, lstm, B, T in [(64, None, 16, 600), (64, None, 64, 1000), (80, None, 37, 611), (64, {"hidden_size": 64}, 61, 509),
                      (60, {"bidirectional": False}, 130, 300), (64, None, 256, 1000)]:
    m = uvad_amd.PyanNet2(lstm=lstm, encoding_dim=F); m.build(); seed_weights(m, 1234, 4.0); m = m.to(dev).eval()
    rt = m.runtime(dev)
    g = torch.Generator(device=dev); g.manual_seed(11)
    feats = torch.randn(B, T, F, generator=g, device=dev) * 4.0 - 8.0
    rt.set_gemm_mode("f16p_stream")
    want = rt.classify(feats, want_probs=False)[0].clone()
    torch.cuda.synchronize()
    print("stream kernel done", flush=True)
    rt.set_gemm_mode("f16p")
    t0 = time.time()
    got = rt.classify(feats, want_probs=False)[0]
    torch.cuda.synchronize()
    d = (got - want).abs().max().item()
    ctr = rt._ws[-8192:].view(torch.int32).view(64, 32)[:, 0].cpu().tolist()
    print(f"F={F} lstm={lstm} B={B} T={T}: max |diff| {d:.3e} identical={torch.equal(got, want)} ({time.time() - t0:.2f} s) counters {ctr}", flush=True)
    rt.close()
print("WS_CHECK_DONE")
