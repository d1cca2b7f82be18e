#!/usr/bin/env python3
"""Where does the logit error come from?  GPU (both GEMM modes) and the fp32 torch-CPU reference path against a float64
evaluation of the same network, on IDENTICAL features, at BASELINE cfg-2 scale.  Run on the GPU box.

    python tools/err_probe.py [--batch 256] [--scales 4,2,1] [--seeds 42,43] [--lib path/to/other/libuvad.so]

--lib loads another build of the library (e.g. one compiled with -DUVAD_FAST_GATES, the round-1 gate functions) so that
two builds can be compared on one box; it patches the binding's path for this process only."""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--scales", default="4,2,1")
ap.add_argument("--seeds", default="42")
ap.add_argument("--lib", default=None)
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--json", default=None)
args = ap.parse_args()

import uvad_amd
from uvad_amd import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from uvad_amd.synth import seed_weights, synth_pcm_device
from oracle import torch_ref as tr, parity_stats as ps

torch.set_num_threads(args.threads)
dev = torch.device("cuda:0")
F, B = 64, args.batch
rows = []
for scale in [float(x) for x in args.scales.split(",")]:
    for seed in [int(x) for x in args.seeds.split(",")]:
        m = uvad_amd.PyanNet2(encoding_dim=F); m.build(); seed_weights(m, 1234, scale)
        m.attach_fbank(uvad_amd.FbankConfig(num_filters=F, window_type="hamming")); m = m.to(dev).eval()
        rt = m.runtime(dev)
        pcm = synth_pcm_device(B, 160000, seed=seed, device=dev)
        feats = rt.fbank(pcm)
        fc = feats.cpu()
        sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        t = time.time(); truth = ps.truth_logits(sd, fc, F); t64 = time.time() - t
        cpu = tr.TorchPyanNet2(F); cpu.load_state_dict(sd)
        t = time.time(); ref = cpu(fc)[0].numpy(); t32 = time.time() - t
        st_cpu = ps.error_stats(ref, truth)
        print(f"scale x{scale:g} seed {seed} B={B}: logits {truth.min():.2f}..{truth.max():.2f}  (torch f64 {t64:.1f} s, f32 {t32:.1f} s)")
        print("  " + ps.fmt("CPU fp32 vs f64      ", st_cpu))
        for mode in ("f16p", "f32"):
            rt.set_gemm_mode(mode)
            g, _ = rt.classify(feats, want_probs=False)
            g = g.cpu().numpy()
            st = ps.error_stats(g, truth)
            st2 = ps.error_stats(g, ref)
            print("  " + ps.fmt(f"GPU {mode:5s} vs f64     ", st) + f"   ratio to CPU: max {st['max']/st_cpu['max']:.2f} rms {st['rms']/st_cpu['rms']:.2f} mean {st['mean']/st_cpu['mean']:.2f}")
            print("  " + ps.fmt(f"GPU {mode:5s} vs CPU fp32", st2))
            rows.append({"scale": scale, "seed": seed, "mode": mode, "gpu_vs_f64": st, "cpu_vs_f64": st_cpu, "gpu_vs_cpu": st2})
        rt.close()
if args.json:
    json.dump({"lib": args.lib or "default", "batch": B, "rows": rows}, open(args.json, "w"), indent=1)
