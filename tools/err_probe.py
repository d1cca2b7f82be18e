#!/usr/bin/env python3
"""Where does the logit error come from?  GPU vs fp32 torch-CPU reference vs the double-accumulating C oracle
on bench-like inputs (identical features for all three).  Run on the GPU box."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uvad_amd
from uvad_amd.synth import seed_weights, synth_pcm_device
from oracle import c_oracle as co, torch_ref as tr

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
F = 64
m = uvad_amd.PyanNet2(encoding_dim=F); m.build(); seed_weights(m, 1234, scale)
m.attach_fbank(uvad_amd.FbankConfig(num_filters=F, window_type="hamming")); m = m.to(dev).eval()
rt = m.runtime(dev)
pcm = synth_pcm_device(B, 160000, seed=42, device=dev)
feats = rt.fbank(pcm)
gls = {}
for mode in ("f32", "bf16x6", "f16x3"):
    rt.set_gemm_mode(mode)
    g_, _ = rt.classify(feats, want_probs=False)
    gls[mode] = g_.cpu().numpy()
gl = gls["f16x3"]
cpu = tr.TorchPyanNet2(F); cpu.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
fc = feats.cpu()
rl = cpu(fc)[0].numpy()
sd = {k: v.cpu().numpy() for k, v in m.state_dict().items()}
ol, _ = co.classify(sd, co.ModelCfg(F, 128, 4, 1, 128, 2, 0.01), fc.numpy())
print(f"B={B} scale={scale}: logits range {ol.min():.2f}..{ol.max():.2f}")
print(f"  |GPU - oracle(f64 acc)| = {np.abs(gl-ol).max():.2e}   |CPU fp32 - oracle| = {np.abs(rl-ol).max():.2e}   |GPU - CPU fp32| = {np.abs(gl-rl).max():.2e}")
for mode, g_ in gls.items():
    print(f"  gemm {mode:7s}: |GPU - oracle| max {np.abs(g_-ol).max():.2e} mean {np.abs(g_-ol).mean():.2e}   |GPU - CPU fp32| max {np.abs(g_-rl).max():.2e} mean {np.abs(g_-rl).mean():.2e}")
print(f"  mean abs: GPU-oracle {np.abs(gl-ol).mean():.2e}  CPU-oracle {np.abs(rl-ol).mean():.2e}")
